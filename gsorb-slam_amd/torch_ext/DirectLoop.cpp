// DirectLoop.cpp — SlamLoop's iterations as fixed sequences of C-ABI launches on a persistent workspace (LoopConfig::direct).
//
// Through libtorch autograd (SlamLoop.cpp) a mapping iteration at 1 M Gaussians, 1200x680 is ~60 launches and as many host-side
// operator dispatches around the rasterizer pair: activations and their backward nodes, gradient accumulation, zero fills, five
// Adam launches, allocations of every intermediate — 1.08-1.30 ms per iteration of which the rasterizer pair is 0.55
// (profiles/r04_loop.md). Here nothing inside an iteration goes through a tensor library: libtorch owns the memory (the
// parameters, the Adam moments, one workspace that lives as long as the map's size), and an iteration is
//   mapping : gsr_forward_ws (fused pair; camera transform, activations and the regularisers' partial sums inside its projection kernel: pre_Tcw, raw) -> gsr_map_loss_forward -> gsr_map_loss_finish -> gsr_map_loss_backward ->
//             gsr_backward [with gsr_map_update fused into its per-splat stage]                                           (13 launches)
//   tracking: gsr_forward_ws [the camera transform inside its projection kernel: pre_Tcw] -> gsr_track_loss -> gsr_backward ->
//             gsr_pose_grad -> gsr_pose_update                                                                          (13 launches)
// with no allocation, no gradient tensor of a raw parameter, and no host synchronisation except where the reference has one
// (the tracking loop reads its loss every iteration, Render.cc:1107 — posted by the pose kernel into host-mapped memory the loop spins on; the mapping loop of RenderForFrame never does: MapFrame
// reads all losses back once, at the end). Reference: src/Render.cc:420-483, :1054-1126; src/Gaussian.cc:97-175.
//
// SHARDED (SetShard; multi-GPU scheme B): every rank holds a shard of the map — one cell of a k-d partition — with its Adam
// moments and rasterizes only that shard; between the forward and the loss sit the compositor of csrc/gsr_shard.h and its collectives, issued
// from here on the loop's stream through the c10d::ProcessGroup the caller hands over (backend "nccl" = RCCL over xGMI; "gloo" in the
// two-processes-on-one-GPU test: staged through the host):
//   gsr_shard_order (the cells front to back for this pose) -> gsr_forward_ws (own shard) -> ALL-GATHER (silhouette, surface depth: 2 planes) ->
//   gsr_composite_forward -> ALL-REDUCE (the four premultiplied planes) -> the loss kernels on the composite [mapping: ALL-REDUCE of the three
//   regulariser sums before gsr_map_loss_finish] -> gsr_composite_backward_local -> ALL-GATHER (g . L: 1 plane) -> gsr_composite_backward_occlusion
//   -> gsr_backward on the layer's gradient [mapping: with the Adam step fused; tracking: its per-splat stage adds the cell's pose sums to 64 accumulator rows
//   -> ALL-REDUCE of the rows (3 KB) -> gsr_pose_finish]. No per-splat data crosses ranks: north_star's "all-reduce on pose / loss gradients only".
#include "SlamLoop.h"

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>
#include <rccl/rccl.h> // (types and prototypes only: the library is opened at run time, below)
#include <dlfcn.h>

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>

#include "../../include/gsr.h"

namespace ORB_SLAM2 {

namespace {
void chk(int rc, const char* what)
{
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + gsr_error_string(rc));
}
float* f(const torch::Tensor& t) { return t.data_ptr<float>(); }
char* b(const torch::Tensor& t) { return reinterpret_cast<char*>(t.data_ptr()); }
} // namespace

// Everything an iteration touches besides the parameters and their moments. Sized by the map (n) and the image; rebuilt when
// the map grows or shrinks (AddGaussians / PruneLowOpacity replace the parameter tensors anyway).
struct SlamLoop::Direct {
    int64_t n = -1;
    size_t binning_bytes = 0;
    torch::Tensor geom, image, binning;                                  // the rasterizer's three workspaces
    torch::Tensor mc, opac, scales, rots, radii;                         // what the rasterizer takes
    torch::Tensor layers;                                                // [6,H,W] what it renders: rgb, depth, silhouette, surface depth — views below
    torch::Tensor out_color, out_sur, out_ds;
    torch::Tensor G;                                                     // [5,H,W] the loss's gradient: d/d rgb, d/d depth, d/d silhouette (zero) — views below
    torch::Tensor g_image, g_ds, g_ssim, dmaps, ssim_partial, partial6;  // upstream gradients of the renders, SSIM / loss scratch
    torch::Tensor d_mc, d_m2d, d_col, d_opac, d_scale, d_rot;            // the rasterizer's gradients
    torch::Tensor loss_partial, sums, reg_partial, reg_out, neg_c, Tcw, bg, view, proj, campos, history;
    torch::Tensor pose, pose_moments, best, pose_partial;                // tracking: [7], [14], [8], [GSR_POSE_PARTIALS, 12]
    torch::Tensor pose_acc;                                              // [64 * 12 + 4] the fused pose step's accumulator rows (zero between launches); word 768: the sharded loop's
                                                                         // overflow flag, which travels in the rows' all-reduce (every rank skips, or none)
    torch::Tensor pose_partial_buf;                                      // [GSR_POSE_PARTIALS * 12 + 4] pose_partial's storage with the same flag word behind it
    torch::Tensor tickets;                                               // [2 * GSR_TICKET_WORDS] arrival counters of the kernels that finish their own sums (zero between launches)
    int64_t history_len = 0;
    // The tracking loop looks at every loss (Render.cc:1107). Read back with a copy and a stream synchronisation that look costs ~20 us per
    // iteration (a 4 us copy kernel, the interrupt that wakes the host, the next launch's latency: the GPU idles through all three); the loop's
    // losses are instead POSTED by the pose kernel into host memory the device can write (fine-grained, mapped) and the host spins on the slot.
    float* posted = nullptr;        // host address [posted_len]
    float* posted_dev = nullptr;    // the device's address of the same words
    int64_t posted_len = 0;
    bool fresh = true; // the workspace was (re)built for a new map size: one pre-flight forward sizes the binning workspace before the first batch of iterations
    // sharded (SetShard): the composite of all ranks' layers and what its backward needs
    c10::intrusive_ptr<c10d::ProcessGroup> pg;
    int rank = 0, world = 1;
    torch::Tensor last_sums;                                             // [12] the pose sums of the sharded loop's last tracking step
    bool order_stale = true;                                             // the pose changed since gsr_shard_order ran
    bool staged = false;                                                 // the group cannot move device tensors (gloo): through the host
    ncclComm_t comm = nullptr;                                           // backend "nccl": the loop's OWN RCCL communicator — its collectives are enqueued on the loop's
                                                                         // stream like any kernel (the process group's run on its internal stream behind two event hops
                                                                         // and ~25 us of host work each: 5 per mapping iteration); the group carries the bootstrap only
    hipStream_t stream = nullptr;
    torch::Tensor kd_nodes, order;                                       // [world - 1, 4] the partition (gsr_shard_order), [world] int64 the ranks front to back
    torch::Tensor gathered, comp, D, c_own, c_all, reg_tot;              // [world,2,H,W]; [6,H,W] composite rgb, depth, silhouette, surface depth;
                                                                         // [5,H,W] d/d own layer (4) and d/d own silhouette; [H,W]; [world,H,W]; [4]
    // the BAND exchange (LoopConfig::band_exchange): this rank's band of pixel rows [b0, b1), every rank's layer on it, every rank's layer gradient on it
    struct Msg { int peer; float* send; float* recv; size_t n; };         // one contiguous block of rows of one plane: to a peer (send) OR from it (recv)
    bool band = false;
    int b0 = 0, b1 = 0;
    torch::Tensor L_all, D_all, rows_all, frame_sums;                    // [world,6,H,W]; [world,5,H,W]; [world,16] {sums[8], reg_out[4], loss slot, 3 unused}; [8] (word 2: the frame's valid depth pixels)
    std::vector<Msg> fwd_map, fwd_track, fwd_render, bwd_map, bwd_track, bwd_render; // the exchanges' messages (built with the workspace: pointers and counts only)
    void p2p(const std::vector<Msg>& ms);                                // one grouped exchange on the loop's stream
    ~Direct();
    bool sharded() const { return world > 1 || pg; }
    void all_gather(torch::Tensor out, const torch::Tensor& in);         // out [world, ...] <- every rank's `in`
    void all_reduce(torch::Tensor t);                                    // in-place sum
};

namespace {
// RCCL is resolved when the first sharded loop asks for it, from the library the process group's backend has already loaded
// (torch/lib/librccl.so): programs that never shard never load it (linked in at build time its exit-time teardown raced libtorch's).
struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    static const Rccl& get()
    {
        static const Rccl r = [] {
            Rccl x;
            void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) throw std::runtime_error(std::string("the sharded loop needs librccl.so: ") + dlerror());
            auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) throw std::runtime_error(std::string("librccl.so lacks ") + n); return p; };
            x.GetUniqueId = (decltype(x.GetUniqueId))sym("ncclGetUniqueId"); x.CommInitRank = (decltype(x.CommInitRank))sym("ncclCommInitRank");
            x.CommDestroy = (decltype(x.CommDestroy))sym("ncclCommDestroy"); x.AllGather = (decltype(x.AllGather))sym("ncclAllGather");
            x.AllReduce = (decltype(x.AllReduce))sym("ncclAllReduce"); x.GetErrorString = (decltype(x.GetErrorString))sym("ncclGetErrorString");
            x.Send = (decltype(x.Send))sym("ncclSend"); x.Recv = (decltype(x.Recv))sym("ncclRecv");
            x.GroupStart = (decltype(x.GroupStart))sym("ncclGroupStart"); x.GroupEnd = (decltype(x.GroupEnd))sym("ncclGroupEnd");
            return x;
        }();
        return r;
    }
};
void nccl_chk(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + Rccl::get().GetErrorString(r));
}
} // namespace

SlamLoop::Direct::~Direct()
{
    if (comm) (void)Rccl::get().CommDestroy(comm);
    if (posted) (void)hipHostFree(posted);
}

namespace {
constexpr uint32_t kNotPosted = 0xFFFFFFFFu; // (a NaN no kernel produces: the pose kernel's own NaN is the canonical one)
// the loss the pose kernel of this iteration posts: spin on the word; every few thousand looks ask the stream whether it has run dry
// (a kernel that died never posts) and give up after ten seconds
float wait_posted(const float* slot, hipStream_t st)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1;; spins++) {
        uint32_t v;
        v = __atomic_load_n(reinterpret_cast<const uint32_t*>(slot), __ATOMIC_ACQUIRE);
        if (v != kNotPosted) { float x; std::memcpy(&x, &v, 4); return x; }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spins & 0xFFFu) == 0) {
            const hipError_t q = hipStreamQuery(st);
            if (q != hipErrorNotReady) { // drained (or failed): one last look
                v = __atomic_load_n(reinterpret_cast<const uint32_t*>(slot), __ATOMIC_ACQUIRE);
                if (v != kNotPosted) { float x; std::memcpy(&x, &v, 4); return x; }
                throw std::runtime_error(std::string("tracking loop: the pose kernel never posted its loss (") + hipGetErrorString(q) + ")");
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) throw std::runtime_error("tracking loop: no loss posted within 10 s");
        }
    }
}
} // namespace

void SlamLoop::Direct::all_gather(torch::Tensor out, const torch::Tensor& in)
{
    if (!pg) { out.select(0, 0).copy_(in); return; }
    if (comm) {
        const auto src = in.contiguous();
        nccl_chk(Rccl::get().AllGather(src.data_ptr<float>(), out.data_ptr<float>(), (size_t)src.numel(), ncclFloat, comm, stream), "ncclAllGather");
        return;
    }
    if (staged) {
        auto hin = in.to(torch::kCPU).contiguous().reshape({-1}), hout = torch::empty({out.numel()}, hin.options()); // (flat: the backends chunk the output by rank)
        pg->_allgather_base(hout, hin)->wait();
        out.copy_(hout.reshape(out.sizes()));
        return;
    }
    auto src = in.contiguous().reshape({-1}), dst = out.reshape({-1}); // (views: `out` is contiguous)
    pg->_allgather_base(dst, src)->wait(); // (RCCL: enqueued behind this stream's work; wait() orders the stream, not the host)
}

void SlamLoop::Direct::all_reduce(torch::Tensor t)
{
    if (!pg) return;
    if (comm) {
        nccl_chk(Rccl::get().AllReduce(t.data_ptr<float>(), t.data_ptr<float>(), (size_t)t.numel(), ncclFloat, ncclSum, comm, stream), "ncclAllReduce");
        return;
    }
    if (staged) {
        auto h = t.to(torch::kCPU).contiguous();
        std::vector<torch::Tensor> v{h};
        pg->allreduce(v)->wait();
        t.copy_(h);
        return;
    }
    std::vector<torch::Tensor> v{t};
    pg->allreduce(v)->wait();
}

// One grouped point-to-point exchange: every message is a contiguous block of floats to one peer and one from it. RCCL: ncclSend / ncclRecv inside one
// group = one launch on the loop's stream, no packing. Any other transport (the process group's own collectives; gloo staged through the host): the
// blocks of a peer are packed in message order and travel through alltoall_base — both ends enumerate a pair's blocks in the same order.
void SlamLoop::Direct::p2p(const std::vector<Msg>& ms)
{
    if (!pg || ms.empty()) return;
    if (comm) {
        const auto& r = Rccl::get();
        nccl_chk(r.GroupStart(), "ncclGroupStart");
        for (const auto& m : ms) {
            if (m.send) nccl_chk(r.Send(m.send, m.n, ncclFloat, m.peer, comm, stream), "ncclSend");
            else nccl_chk(r.Recv(m.recv, m.n, ncclFloat, m.peer, comm, stream), "ncclRecv");
        }
        nccl_chk(r.GroupEnd(), "ncclGroupEnd");
        return;
    }
    std::vector<int64_t> s_split(world, 0), r_split(world, 0), s_off(world, 0), r_off(world, 0);
    for (const auto& m : ms) (m.send ? s_split : r_split)[m.peer] += (int64_t)m.n;
    int64_t s_tot = 0, r_tot = 0;
    for (int k = 0; k < world; k++) { s_off[k] = s_tot; s_tot += s_split[k]; r_off[k] = r_tot; r_tot += r_split[k]; }
    const auto fo = torch::TensorOptions().device(rows_all.device()).dtype(torch::kFloat32);
    auto view = [&](float* p, size_t n) { return torch::from_blob(p, {(int64_t)n}, fo); };
    auto in = torch::empty({s_tot}, fo), out = torch::empty({r_tot}, fo);
    for (const auto& m : ms) if (m.send) { in.narrow(0, s_off[m.peer], (int64_t)m.n).copy_(view(m.send, m.n)); s_off[m.peer] += (int64_t)m.n; }
    if (staged) {
        auto hin = in.to(torch::kCPU), hout = torch::empty({r_tot}, hin.options());
        pg->alltoall_base(hout, hin, r_split, s_split)->wait();
        out.copy_(hout);
    } else
        pg->alltoall_base(out, in, r_split, s_split)->wait();
    for (const auto& m : ms) if (!m.send) { view(m.recv, m.n).copy_(out.narrow(0, r_off[m.peer], (int64_t)m.n)); r_off[m.peer] += (int64_t)m.n; }
}

void SlamLoop::SetShard(c10::intrusive_ptr<c10d::ProcessGroup> pg, int rank, int world, const torch::Tensor& kd_nodes)
{
    if (!direct_()) throw std::runtime_error("SetShard needs LoopConfig::direct");
    if (world < 1 || rank < 0 || rank >= world || world > 32) throw std::runtime_error("SetShard: bad rank / world");
    if (pg && (pg->getRank() != rank || pg->getSize() != world)) throw std::runtime_error("SetShard: the group's rank / size differ");
    if (!pg && world > 1) throw std::runtime_error("SetShard: world > 1 needs a process group");
    if (kd_nodes.defined() && kd_nodes.numel() > 0 && (kd_nodes.dim() != 2 || kd_nodes.size(0) != world - 1 || kd_nodes.size(1) != 4))
        throw std::runtime_error("SetShard: kd_nodes must be [world - 1, 4]");
    if (!d_) d_ = std::make_shared<Direct>();
    Direct& d = *d_;
    d.pg = pg; d.rank = rank; d.world = world;
    d.staged = pg && pg->getBackendName() != "nccl";
    if (d.comm) { (void)Rccl::get().CommDestroy(d.comm); d.comm = nullptr; }
    d.fresh = true; // (the first ShardRenderStep / batch on this partition sizes the workspace)
    if (pg && !d.staged) { // the communicator's id travels through the group (a device tensor: the group's backend is RCCL too)
        c10::DeviceGuard guard(dev_);
        // Every rank must end up on the SAME transport: a rank whose own communicator failed says so in a second broadcast-free exchange (an
        // all-reduce of one flag through the group) and all of them fall back to the group's own collectives (c10d's RCCL communicator: the same
        // wire, issued through libtorch on this stream — slower to launch, never wrong).
        int ok = 1;
        std::string why;
        //  the sequence of group collectives is the same on every path: a rank that fails BEFORE the broadcast (librccl not loadable,
        // ncclGetUniqueId) still takes part in it — with a zeroed id — and says so in the flag all-reduce behind it
        ncclUniqueId id;
        std::memset(&id, 0, sizeof(id));
        try {
            (void)Rccl::get();
            if (rank == 0) nccl_chk(Rccl::get().GetUniqueId(&id), "ncclGetUniqueId");
        } catch (const std::exception& e) {
            ok = 0; why = e.what();
            std::memset(&id, 0, sizeof(id));
        }
        {
            auto bytes = torch::empty({(int64_t)sizeof(id) + 1}, torch::kUInt8); // (the id and rank 0's "I have one" byte)
            std::memcpy(bytes.data_ptr(), &id, sizeof(id));
            bytes.data_ptr<uint8_t>()[sizeof(id)] = (uint8_t)ok;
            auto on_dev = bytes.to(dev_);
            std::vector<torch::Tensor> v{on_dev};
            pg->broadcast(v)->wait();
            bytes = on_dev.to(torch::kCPU);
            std::memcpy(&id, bytes.data_ptr(), sizeof(id));
            if (!bytes.data_ptr<uint8_t>()[sizeof(id)] && ok) { ok = 0; why = "rank 0 has no RCCL id"; }
        }
        if (ok) {
            try {
                nccl_chk(Rccl::get().CommInitRank(&d.comm, world, id, rank), "ncclCommInitRank");
            } catch (const std::exception& e) {
                ok = 0; why = e.what();
                d.comm = nullptr;
            }
        }
        auto flag = torch::full({1}, (float)ok, torch::TensorOptions().device(dev_).dtype(torch::kFloat32));
        std::vector<torch::Tensor> fv{flag};
        c10d::AllreduceOptions opts;
        opts.reduceOp = c10d::ReduceOp::MIN;
        pg->allreduce(fv, opts)->wait();
        if (flag.item<float>() < 0.5f) {
            if (d.comm) { (void)Rccl::get().CommDestroy(d.comm); d.comm = nullptr; }
            if (!ok) fprintf(stderr, "[gsr] SetShard: own RCCL communicator unavailable on rank %d (%s): the loop's collectives go through the process group\n", rank, why.c_str());
        }
    }
    d.kd_nodes = (kd_nodes.defined() && kd_nodes.numel() > 0) ? kd_nodes.to(dev_, torch::kFloat32).contiguous() : torch::Tensor();
    d.order = torch::arange(world, torch::TensorOptions().device(dev_).dtype(torch::kInt64)); // (no partition given: the ranks ARE the order, e.g. depth slabs)
    shard_ = true;
    d.gathered = torch::Tensor(); // (re-sized by ensure_direct_)
}

// which way the sharded loop's collectives travel: "rccl" (the loop's own communicator, launched on the loop's stream), "c10d" (the process group's
// collectives), "staged" (a host-side backend: device -> host -> group -> device), "local" (one rank, no group), "" (not sharded)
std::string SlamLoop::ShardTransport() const
{
    if (!shard_ || !d_) return "";
    if (!d_->pg) return "local";
    return d_->comm ? "rccl" : d_->staged ? "staged" : "c10d";
}

SlamLoop::~SlamLoop() = default;

void* SlamLoop::stream_() const { return (void*)c10::hip::getCurrentHIPStream(dev_.index()).stream(); }

void SlamLoop::ensure_direct_(int64_t history_len)
{
    if (!cfg_.fused_pair || !cfg_.fused_ops) throw std::runtime_error("LoopConfig::direct needs fused_pair and fused_ops");
    if (!d_) d_ = std::make_shared<Direct>();
    Direct& d = *d_;
    const auto fo = torch::TensorOptions().device(dev_).dtype(torch::kFloat32);
    const auto bo = torch::TensorOptions().device(dev_).dtype(torch::kByte);
    const int64_t n = size();
    if (!d.image.defined()) { // per image size: once
        const int64_t np = (int64_t)gsr_ssim_partials(3, H_, W_);
        d.image = torch::empty({(int64_t)gsr_image_bytes(W_, H_)}, bo);
        d.layers = torch::empty({6, H_, W_}, fo); // (one block: the compositor takes (rgb, depth) as four consecutive planes, the all-gather (silhouette, surface depth) as two)
        d.out_color = d.layers.slice(0, 0, 3); d.out_ds = d.layers.slice(0, 3, 5); d.out_sur = d.layers.slice(0, 5, 6);
        d.G = torch::zeros({5, H_, W_}, fo);
        d.g_image = d.G.slice(0, 0, 3); d.g_ds = d.G.slice(0, 3, 5); // (the silhouette plane's gradient stays zero: its mask is detached)
        d.g_ssim = torch::empty({3, H_, W_}, fo); d.dmaps = torch::empty({3, 3, H_, W_}, fo); d.ssim_partial = torch::empty({np}, fo); d.partial6 = torch::empty({np * 6}, fo);
        d.loss_partial = torch::empty({GSR_LOSS_PARTIALS * 5}, fo); d.sums = torch::empty({8}, fo); d.reg_out = torch::empty({4}, fo);
        d.neg_c = torch::full({1}, -(cfg_.im_weight_mapping * (1 - cfg_.lam)), fo);
        d.Tcw = torch::eye(4, fo); d.bg = torch::zeros({3}, fo); d.view = torch::eye(4, fo); d.campos = torch::zeros({3}, fo);
        d.proj = rasterizer_.raster_settings_.projmatrix.to(dev_, torch::kFloat32).contiguous();
        d.pose = torch::zeros({7}, fo); d.pose_moments = torch::zeros({14}, fo); d.best = torch::zeros({8}, fo);
        d.tickets = torch::zeros({2 * GSR_TICKET_WORDS}, fo.dtype(torch::kInt32));
        d.pose_acc = torch::zeros({64 * 12 + 16}, fo); d.last_sums = torch::zeros({12}, fo); // (word 768: the overflow flag; words 776..783: the band exchange's tracking-loss sums — they travel in the rows' all-reduce)
        d.frame_sums = torch::zeros({8}, fo);
    }
    if (d.n != n) { // per map size
        d.n = n;
        d.fresh = true;
        d.geom = torch::empty({(int64_t)gsr_geom_bytes((int)n)}, bo);
        //  the header {num_rendered, overflow, ...} is written by the forward's kernels — an EMPTY shard never launches one, and the loss / pose
        // kernels still read the overflow flag through it: it starts out as "nothing rendered, nothing overflowed"
        d.geom.slice(0, 0, std::min<int64_t>(256, d.geom.numel())).zero_();
        d.mc = torch::empty({n, 3}, fo); d.opac = torch::empty({n}, fo); d.scales = torch::empty({n, 3}, fo); d.rots = torch::empty({n, 4}, fo);
        d.radii = torch::empty({n}, fo.dtype(torch::kInt32));
        d.d_mc = torch::empty({n, 3}, fo); d.d_m2d = torch::empty({n, 3}, fo); d.d_col = torch::empty({n, 3}, fo); d.d_opac = torch::empty({n}, fo);
        d.d_scale = torch::empty({n, 3}, fo); d.d_rot = torch::empty({n, 4}, fo);
        d.reg_partial = torch::empty({3 * ((n + 255) / 256) + 3}, fo);
        d.pose_partial_buf = torch::zeros({GSR_POSE_PARTIALS * 12 + 4}, fo);
        d.pose_partial = d.pose_partial_buf.slice(0, 0, GSR_POSE_PARTIALS * 12).view({GSR_POSE_PARTIALS, 12});
        d.binning = torch::Tensor(); d.binning_bytes = 0;
        grow_binning_(cfg_.binning_capacity > 0 ? (size_t)cfg_.binning_capacity : 4 * (size_t)n + 65536); // grows at the first synchronised look at an overflow
    }
    if (d.history_len < history_len) { d.history = torch::empty({history_len}, fo); d.history_len = history_len; }
    if (d.posted_len < history_len) {
        if (d.posted) { (void)hipHostFree(d.posted); d.posted = nullptr; }
        void* dp = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&d.posted), (size_t)history_len * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer(&dp, d.posted, 0) != hipSuccess)
            throw std::runtime_error("direct loop: no host-mapped memory for the posted losses");
        d.posted_dev = static_cast<float*>(dp);
        d.posted_len = history_len;
    }
    if (shard_ && !d.gathered.defined()) {
        // comp: the four planes the all-reduce sums, the three regulariser sums right behind them (ONE all-reduce carries both), then the
        // composite's silhouette and surface depth
        const int64_t HW = (int64_t)H_ * W_;
        d.gathered = torch::empty({d.world, 2, H_, W_}, fo); d.comp = torch::zeros({6 * HW + 4}, fo); d.D = torch::empty({5, H_, W_}, fo);
        d.reg_tot = d.comp.slice(0, 4 * HW, 4 * HW + 4);
        d.c_own = torch::empty({1, H_, W_}, fo); d.c_all = torch::empty({d.world, 1, H_, W_}, fo);
        // the band exchange: this rank's band, the buffers and the two exchanges' messages (forward: the rows of MY layer every peer's band needs — with
        // ten rows either side of the colour and silhouette planes for the mapping loss's SSIM window —, backward: every peer's layer gradient on my band)
        const int hb = (H_ + d.world - 1) / d.world;
        auto band_of = [&](int k, int halo, int& lo, int& hi) { lo = std::max(0, std::min(H_, k * hb) - halo); hi = std::min(H_, std::min(H_, (k + 1) * hb) + halo); if (k * hb >= H_) lo = hi = 0; };
        band_of(d.rank, 0, d.b0, d.b1);
        d.band = cfg_.band_exchange && cfg_.fused_loss && cfg_.fused_update && (d.world - 1) * hb < H_; // (every rank a non-empty band: more ranks than rows keep the replicated composite)
        d.fwd_map.clear(); d.fwd_track.clear(); d.fwd_render.clear(); d.bwd_map.clear(); d.bwd_track.clear(); d.bwd_render.clear();
        d.rows_all = torch::zeros({d.world, 16}, fo);
        if (d.band && d.world > 1) {
            const size_t HW = (size_t)H_ * W_;
            d.L_all = torch::zeros({d.world, 6, H_, W_}, fo); d.D_all = torch::zeros({d.world, 5, H_, W_}, fo);
            for (int k = 0; k < d.world; k++) {
                if (k == d.rank) continue;
                for (int halo : {10, 0}) {
                    for (int pl : {0, 1, 2, 4, 3, 5}) {
                        const int h = (pl == 3 || pl == 5) ? 0 : halo; // (depth and surface depth: the band itself)
                        int s0, s1, r0, r1;
                        band_of(k, h, s0, s1); band_of(d.rank, h, r0, r1);
                        // what I send is rows [s0, s1) of my layer (peer k's band); what I receive is rows [r0, r1) of k's layer (my band).
                        // (a message is a send OR a receive: the two bands of a pair may differ in height)
                        // (a tracking iteration on the surface depth reads nothing of the blended depth: plane 3 stays at home, forwards and backwards — the same
                        // LoopConfig on every rank, like everything else of the exchange; ShardRenderStep's gradient has all five planes)
                        const bool track_too = !(pl == 3 && cfg_.use_sur_depth);
                        auto put = [&](const Direct::Msg& m) { if (halo) d.fwd_map.push_back(m); else { d.fwd_render.push_back(m); if (track_too) d.fwd_track.push_back(m); } };
                        if (s1 > s0) put(Direct::Msg{k, f(d.layers) + pl * HW + (size_t)s0 * W_, nullptr, (size_t)(s1 - s0) * W_});
                        if (r1 > r0) put(Direct::Msg{k, nullptr, f(d.L_all) + ((size_t)k * 6 + pl) * HW + (size_t)r0 * W_, (size_t)(r1 - r0) * W_});
                    }
                }
                int k0, k1;
                band_of(k, 0, k0, k1);
                for (int pl = 0; pl < 5; pl++) {
                    const bool track_too = !(pl == 3 && cfg_.use_sur_depth);
                    auto put = [&](const Direct::Msg& m) { d.bwd_map.push_back(m); d.bwd_render.push_back(m); if (track_too) d.bwd_track.push_back(m); };
                    if (d.b1 > d.b0) put(Direct::Msg{k, f(d.D_all) + ((size_t)k * 5 + pl) * HW + (size_t)d.b0 * W_, nullptr, (size_t)(d.b1 - d.b0) * W_});
                    if (k1 > k0) put(Direct::Msg{k, nullptr, f(d.D) + pl * HW + (size_t)k0 * W_, (size_t)(k1 - k0) * W_});
                }
                d.bwd_map.push_back(Direct::Msg{k, f(d.rows_all) + (size_t)d.rank * 16, nullptr, 16});
                d.bwd_map.push_back(Direct::Msg{k, nullptr, f(d.rows_all) + (size_t)k * 16, 16});
            }
        }
    }
}

// (the collectives of an iteration go to the stream its kernels go to)
// Sharded: the composite of all ranks' layers from this rank's render (layers) — comp = {rgb, depth, silhouette, surface depth} of the whole map.
void SlamLoop::shard_composite_forward_(bool pose_moved, bool with_reg)
{
    Direct& d = *d_;
    void* const st = stream_();
    d.stream = (hipStream_t)st;
    const size_t HW = (size_t)H_ * W_;
    // (the cells' order depends on the pose alone: once per MapFrame call, every iteration while the pose is tracked)
    if (pose_moved && d.kd_nodes.defined()) chk(gsr_shard_order(d.world, f(d.kd_nodes), f(d.Tcw), (long long*)d.order.data_ptr<int64_t>(), st), "gsr_shard_order");
    d.all_gather(d.gathered, d.layers.slice(0, 4, 6));
    chk(gsr_composite_forward(d.world, d.rank, (const long long*)d.order.data_ptr<int64_t>(), f(d.gathered), 2, f(d.layers), H_, W_, 1, f(d.comp),
                              f(d.comp) + 4 * HW + 4, f(d.comp) + 5 * HW + 4, st), "gsr_composite_forward");
    // mapping: the regularisers are a sum and a mean over the WHOLE map (Render.cc:449-462) — their three sums ride behind the planes
    d.all_reduce(d.comp.slice(0, 0, (int64_t)(4 * HW + (with_reg ? 3 : 0))));
}

// Sharded: from the loss's gradient on the composite (G: rgb, depth; the silhouette's is zero — a detached mask in both losses) to the gradient
// of this rank's own layer (D: rgb, depth, silhouette): its planes through its prefix transmittance, its silhouette through what it occludes.
void SlamLoop::shard_composite_backward_()
{
    Direct& d = *d_;
    void* const st = stream_();
    d.stream = (hipStream_t)st;
    const long long* const order = (const long long*)d.order.data_ptr<int64_t>();
    chk(gsr_composite_backward_local(d.world, d.rank, order, f(d.gathered), 2, f(d.layers), f(d.G), H_, W_, f(d.D), f(d.c_own), st), "gsr_composite_backward_local");
    d.all_gather(d.c_all, d.c_own);
    chk(gsr_composite_backward_occlusion(d.world, d.rank, order, f(d.gathered), 2, f(d.c_all), nullptr, H_, W_, f(d.D) + (size_t)4 * H_ * W_, st),
        "gsr_composite_backward_occlusion");
}

// ---- the BAND exchange (LoopConfig::band_exchange): what replaces the two functions above in the loops' iterations
bool SlamLoop::band_() const { return shard_ && d_ && d_->band; }

// Every rank's layer on this rank's band of rows (one grouped exchange), then the composite of the band: comp = {rgb, depth, silhouette, surface depth} on
// rows [b0, b1) — the mapping loss's SSIM window reads the colour planes ten rows beyond, so those rows travel and are composited too.
void SlamLoop::band_forward_(bool pose_moved, int kind)
{
    Direct& d = *d_;
    void* const st = stream_();
    d.stream = (hipStream_t)st;
    const size_t HW = (size_t)H_ * W_;
    if (pose_moved && d.kd_nodes.defined()) chk(gsr_shard_order(d.world, f(d.kd_nodes), f(d.Tcw), (long long*)d.order.data_ptr<int64_t>(), st), "gsr_shard_order");
    const bool tracking = kind != kBandMap;
    d.p2p(kind == kBandMap ? d.fwd_map : kind == kBandTrack ? d.fwd_track : d.fwd_render);
    if (d.b1 > d.b0)
        chk(gsr_band_composite_forward(d.world, d.rank, (const long long*)d.order.data_ptr<int64_t>(), d.L_all.defined() ? f(d.L_all) : nullptr, f(d.layers), H_, W_, d.b0, d.b1,
                                       tracking ? 0 : 10, f(d.comp), f(d.comp) + 4 * HW + 4, f(d.comp) + 5 * HW + 4, st), "gsr_band_composite_forward");
}

// From the loss's gradient on the band (G: rgb, depth) to EVERY rank's layer gradient on the band, and each rank's rows back to it (the second exchange;
// a mapping iteration's sixteen loss / regulariser words per rank ride in it): D = d/d own layer (rgb, depth, silhouette) on the whole frame.
void SlamLoop::band_backward_(int kind, const float* g_sil)
{
    Direct& d = *d_;
    void* const st = stream_();
    d.stream = (hipStream_t)st;
    if (d.b1 > d.b0)
        chk(gsr_band_composite_backward(d.world, d.rank, (const long long*)d.order.data_ptr<int64_t>(), d.L_all.defined() ? f(d.L_all) : nullptr, f(d.layers), f(d.G), g_sil, H_, W_,
                                        d.b0, d.b1, d.D_all.defined() ? f(d.D_all) : nullptr, f(d.D), st), "gsr_band_composite_backward");
    d.p2p(kind == kBandMap ? d.bwd_map : kind == kBandTrack ? d.bwd_track : d.bwd_render);
}

void SlamLoop::grow_binning_(size_t capacity)
{
    Direct& d = *d_;
    const size_t bytes = gsr_binning_bytes(capacity);
    if (bytes <= d.binning_bytes) return;
    d.binning = torch::empty({(int64_t)bytes}, torch::TensorOptions().device(dev_).dtype(torch::kByte));
    d.binning_bytes = bytes;
}

// the fused colour + depth / silhouette pass on the workspace (sync-free), and its backward
void SlamLoop::direct_forward_(bool from_world, bool raw, float reg_limit, bool plain, bool plain_sil)
{
    Direct& d = *d_;
    if (d.n == 0) { d.layers.zero_(); return; } // (an empty shard still takes part in the exchange: its layer is nothing)
    const auto& s = rasterizer_.raster_settings_;
    gsr_forward_args a{};
    a.P = (int)d.n; a.D = 0; a.M = 0;
    a.background = f(d.bg); a.width = W_; a.height = H_;
    // from_world (a tracking iteration): the projection kernel moves the world means into the camera frame of the pose on the device itself and leaves them in
    // d.mc for the backward (gsr_forward_args.pre_Tcw) — gsr_to_camera's launch less per iteration
    a.means3D = from_world ? f(xyz) : f(d.mc); a.pre_Tcw = from_world ? f(d.Tcw) : nullptr; a.means_cam_out = from_world ? f(d.mc) : nullptr;
    // raw (a mapping iteration of the unsharded loop): the projection kernel also applies the activations (gsr_map_prepare's: sigmoid, exp, normalize), leaves them in
    // d.opac / d.scales / d.rots for the backward and writes the scale regularisers' partial sums (gsr_forward_args.raw) — gsr_map_prepare's launch less per iteration
    gsr_raw_outputs ro{f(d.opac), f(d.scales), f(d.rots), reg_limit, f(d.reg_partial)};
    a.colors_precomp = f(rgb); a.opacities = raw ? f(logit_opacities) : f(d.opac); a.scales = raw ? f(log_scales) : f(d.scales); a.scale_modifier = s.scale_modifier;
    a.rotations = raw ? f(unnorm_quat) : f(d.rots); a.raw = raw ? &ro : nullptr; a.viewmatrix = f(d.view); a.projmatrix = f(d.proj); a.cam_pos = f(d.campos);
    a.tan_fovx = s.tanfovx; a.tan_fovy = s.tanfovy; a.prefiltered = 0;
    a.out_color = f(d.out_color); a.out_depth = f(d.out_sur); a.radii = d.radii.data_ptr<int>(); a.out_ds = plain ? nullptr : f(d.out_ds); // (plain: the three colour channels only — a tracking iteration on the surface depth)
    if (plain && plain_sil) a.out_sil = f(d.out_ds) + (size_t)H_ * W_; // (... and the silhouette 1 - T into its plane of the layer: what the sharded compositor needs of the pair)
    chk(gsr_forward_ws(&a, b(d.geom), b(d.binning), d.binning_bytes, b(d.image), stream_()), "gsr_forward_ws");
}

void SlamLoop::direct_backward_(bool detach_depth_colour, bool means_only, const ::gsr_map_update_args* fused, const ::gsr_pose_step_args* pose_step)
{
    Direct& d = *d_;
    if (d.n == 0 && !pose_step) return;
    const auto& s = rasterizer_.raster_settings_;
    gsr_backward_args a{};
    a.P = (int)d.n; a.D = 0; a.M = 0; a.R = -1;
    a.background = f(d.bg); a.width = W_; a.height = H_;
    a.means3D = f(d.mc); a.colors_precomp = f(rgb); a.scales = f(d.scales); a.scale_modifier = s.scale_modifier; a.rotations = f(d.rots);
    a.viewmatrix = f(d.view); a.projmatrix = f(d.proj); a.cam_pos = f(d.campos); a.tan_fovx = s.tanfovx; a.tan_fovy = s.tanfovy;
    a.radii = d.radii.data_ptr<int>();
    a.geom_buffer = b(d.geom); a.binning_buffer = b(d.binning); a.image_buffer = b(d.image);
    a.dL_dpix = f(d.g_image); a.dL_dds = f(d.g_ds); a.ds_detach_depth = detach_depth_colour ? 1 : 0;
    a.dds_depth_only = 1; // (the silhouette is a detached mask in both losses: plane 1 of g_ds would be zeros)
    if (shard_) { // the own layer's gradient comes from the compositor, and the own silhouette has one: what the layer occludes
        a.dL_dpix = f(d.D); a.dL_dds = f(d.D) + (size_t)3 * H_ * W_; a.dds_depth_only = 0;
        // (round 6) tracking on the surface depth: the composite's depth plane carries no gradient, so the layer's is zero too — only the silhouette's plane goes in
        if (means_only && cfg_.use_sur_depth && pose_step && !std::getenv("GSR_EXP_TRACK_DUAL_BWD")) { a.dL_dds = f(d.D) + (size_t)4 * H_ * W_; a.dds_depth_only = 2; }
    } else if (means_only && cfg_.use_sur_depth && !std::getenv("GSR_EXP_TRACK_DUAL_BWD")) { // (A/B hook: the fused pair's no-colour kernel, as in round 5)
        // (round 6) a tracking iteration on the surface (median) depth: the loss has no gradient through the blended depth and the silhouette is a detached mask —
        // nothing arrives through the fused channels (gsr_track_loss wrote zeros): the backward blend runs its plain form without DUAL's depth recursion and without
        // the colour sums (118 registers, four waves per SIMD)
        a.dL_dds = nullptr; a.dds_depth_only = 0;
    }
    a.fused_map_update = fused; // (the per-splat stage then takes the Adam step itself and writes no gradient)
    a.fused_pose_step = pose_step; // (tracking: the per-splat stage forms the pose sums and its last workgroup takes the pose step: no gradient tensor)
    if (!fused && !pose_step) a.dL_dmean3D = f(d.d_mc);
    if (!means_only && !fused) { // (tracking optimises the pose only: the per-splat stage then skips the covariance -> scale / rotation chain and 56 bytes of stores per Gaussian)
        a.dL_dmean2D = f(d.d_m2d); a.dL_dopacity = f(d.d_opac); a.dL_dcolor = f(d.d_col);
        a.dL_dscale = f(d.d_scale); a.dL_drot = f(d.d_rot);
    }
    a.stages = GSR_STAGE_BLEND | GSR_STAGE_SPLAT; // one backward per forward: the forward left the accumulators clear
    chk(gsr_backward(&a, stream_()), "gsr_backward");
}

// After a synchronised look at the workspace: did the last forward fit? If not, the workspace grows to 1.5 x what it needed.
bool SlamLoop::direct_overflowed_()
{
    int R = 0, ov = 0;
    if (d_->n == 0) return false; // (an empty shard renders nothing: nothing to overflow)
    chk(gsr_ws_status(b(d_->geom), stream_(), &R, &ov), "gsr_ws_status");
    if (ov) grow_binning_((size_t)R + (size_t)R / 2 + 65536);
    return ov != 0;
}

void SlamLoop::direct_map_iteration_(const LoopFrame& fr, float* loss_slot)
{
    Direct& d = *d_;
    void* const st = stream_();
    const size_t n = (size_t)d.n;
    const float limit = (float)(0.1 * cfg_.scene_radius), wl = (float)cfg_.reg_long_weight, wsc = (float)cfg_.reg_scalar_weight;
    if (shard_ && !cfg_.fused_loss) throw std::runtime_error("the sharded loop needs LoopConfig::fused_loss");
    // (the unsharded loop with the fused loss: camera transform, activations and the regularisers' partial sums ride in the projection kernel; the other
    // configurations finish the regularisers' sums right here, in gsr_map_prepare's second launch)
    const bool in_projection = cfg_.fused_loss && n > 0;
    if (n == 0) d.reg_tot.zero_();
    else if (!in_projection)
        chk(gsr_map_prepare(n, f(xyz), f(logit_opacities), f(log_scales), f(unnorm_quat), f(d.Tcw), f(d.mc), f(d.opac), f(d.scales), f(d.rots), limit, wl, wsc,
                            f(d.reg_partial), shard_ ? f(d.reg_tot) : cfg_.fused_loss ? nullptr : f(d.reg_out), st), "gsr_map_prepare");
    direct_forward_(in_projection, in_projection, limit);
    // (sharded: the cell's three regulariser sums go into the all-reduce behind the composite's planes — the finish of the rows the projection kernel wrote)
    // (band exchange: nobody all-reduces them — the band's finish kernel adds the rows up itself, like the unsharded loop's: one single-workgroup launch less)
    const bool band = band_();
    if (in_projection && shard_ && !band)
        chk(gsr_map_prepare(n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, limit, wl, wsc, f(d.reg_partial), f(d.reg_tot), st), "gsr_map_prepare (finish)");
    if (band) { band_forward_(d.order_stale, kBandMap); d.order_stale = false; }
    else if (shard_) { shard_composite_forward_(d.order_stale, true); d.order_stale = false; }
    // Render.cc:436-471: lam * L1 + (1 - lam) * (1 - SSIM), masked depth L1, masked surface-depth L1 (no gradient), the regularisers
    const float w3[3] = {(float)(cfg_.im_weight_mapping * cfg_.lam), (float)cfg_.depth_weight_mapping, (float)cfg_.sur_depth_weight_mapping};
    const float c_ssim = (float)(cfg_.im_weight_mapping * (1 - cfg_.lam));
    const size_t HW = (size_t)H_ * W_;
    const float* const im0 = shard_ ? f(d.comp) : f(d.layers); // (both blocks: rgb, depth, silhouette, surface depth; comp keeps four floats between depth and silhouette)
    const size_t gap = shard_ ? 4 : 0;
    const float *img = im0, *dep = im0 + 3 * HW, *sil = im0 + 4 * HW + gap, *sur = im0 + 5 * HW + gap;
    if (band) {
        // the loss on this rank's band of rows: its raw sums (and the cell's regulariser sums, and the loss slot that says whether this rank's forward
        // overflowed) go out with the gradient's exchange; the gradient planes need one total only — the frame's count of valid depth pixels, a constant of the frame
        float* const row = f(d.rows_all) + (size_t)d.rank * 16;
        chk(gsr_map_loss_forward_rows(img, dep, sur, sil, f(fr.rgb), f(fr.depth), H_, W_, taps_host_.data(), 0.99f, f(d.partial6), f(d.dmaps), d.b0, d.b1, st), "gsr_map_loss_forward_rows");
        // (an empty cell has no rows: its regulariser sums are the zeros d.reg_tot holds)
        chk(gsr_map_loss_finish_rows(f(d.partial6), n > 0 ? f(d.reg_partial) : f(d.reg_tot), n > 0 ? n : 1, H_, W_, w3, c_ssim, wl, wsc, b(d.geom), row, row + 8, row + 12, d.b0, d.b1, st),
            "gsr_map_loss_finish_rows");
        chk(gsr_map_loss_backward_rows(img, dep, f(fr.rgb), f(fr.depth), f(d.dmaps), H_, W_, taps_host_.data(), w3, f(d.neg_c), f(d.frame_sums), f(d.g_image), f(d.g_ds), d.b0, d.b1, st),
            "gsr_map_loss_backward_rows");
        band_backward_(kBandMap);
        chk(gsr_shard_map_totals(d.world, f(d.rows_all), H_, W_, w3, c_ssim, wl, wsc, f(d.sums), f(d.reg_out), loss_slot, st), "gsr_shard_map_totals");
    } else if (cfg_.fused_loss) { // two passes over the image and one single-workgroup kernel between them
        chk(gsr_map_loss_forward(img, dep, sur, sil, f(fr.rgb), f(fr.depth), H_, W_, taps_host_.data(), 0.99f, f(d.partial6), f(d.dmaps), st), "gsr_map_loss_forward");
        // (sharded: the regulariser sums of the whole map stand in as ONE row)
        chk(gsr_map_loss_finish(f(d.partial6), shard_ ? f(d.reg_tot) : f(d.reg_partial), shard_ ? 1 : n, H_, W_, w3, c_ssim, wl, wsc, b(d.geom), f(d.sums), f(d.reg_out),
                                loss_slot, st), "gsr_map_loss_finish");
        chk(gsr_map_loss_backward(img, dep, f(fr.rgb), f(fr.depth), f(d.dmaps), H_, W_, taps_host_.data(), w3, f(d.neg_c), f(d.sums), f(d.g_image), f(d.g_ds), st),
            "gsr_map_loss_backward");
    } else {
        chk(gsr_pixel_loss(img, dep, sur, sil, f(fr.rgb), f(fr.depth), H_, W_, 1, 0.99f, w3, f(d.loss_partial), f(d.sums), st), "gsr_pixel_loss");
        chk(gsr_ssim_forward(img, f(fr.rgb), 3, H_, W_, taps_host_.data(), f(d.ssim_partial), f(d.dmaps), st), "gsr_ssim_forward");
        chk(gsr_ssim_backward(img, f(fr.rgb), f(d.dmaps), 3, H_, W_, taps_host_.data(), f(d.neg_c), f(d.g_ssim), st), "gsr_ssim_backward");
        chk(gsr_pixel_loss_backward_add(img, dep, sil, f(fr.rgb), f(fr.depth), H_, W_, 1, 0.99f, w3, f(d.sums), nullptr, f(d.g_ssim), f(d.g_image), f(d.g_ds), st),
            "gsr_pixel_loss_backward_add");
        chk(gsr_map_loss_total(f(d.sums), f(d.ssim_partial), (int)d.ssim_partial.numel(), (size_t)3 * H_ * W_, c_ssim, f(d.reg_out), b(d.geom), loss_slot, st),
            "gsr_map_loss_total");
    }
    ::gsr_map_update_args u{};
    u.n = n; u.xyz = f(xyz); u.rgb = f(rgb); u.unnorm_quat = f(unnorm_quat); u.logit = f(logit_opacities); u.log_scales = f(log_scales);
    for (int g = 0; g < 5; g++) {
        u.exp_avg[g] = f(fopt_->exp_avg(g)); u.exp_avg_sq[g] = f(fopt_->exp_avg_sq(g));
        u.lr[g] = fopt_->lr(g); u.step[g] = fopt_->advance_step(g);
    }
    u.dL_dmeans_cam = f(d.d_mc); u.dL_dcolors = f(d.d_col); u.dL_drotations = f(d.d_rot); u.dL_dopacities = f(d.d_opac); u.dL_dscales = f(d.d_scale);
    u.opacities = f(d.opac); u.scales = f(d.scales); u.Tcw = f(d.Tcw);
    u.reg_out = f(d.reg_out); u.reg_limit = limit; u.w_long = wl; u.w_scalar = wsc;
    u.geom = b(d.geom); u.beta1 = 0.9; u.beta2 = 0.999; u.eps = fopt_->eps();
    if (shard_ && !band) shard_composite_backward_();
    if (n == 0) return;
    if (cfg_.fused_update) {
        direct_backward_(false, false, &u, nullptr); // backward and update in the same per-splat pass (gsr_backward_args.fused_map_update)
    } else {
        direct_backward_(false, false, nullptr, nullptr);
        chk(gsr_map_update(&u, st), "gsr_map_update");
    }
}

std::vector<double> SlamLoop::MapFrame(const LoopFrame& fr, int iters)
{
    std::vector<double> losses;
    if (iters <= 0 || (size() == 0 && !shard_)) return losses;
    if (!direct_()) {
        for (int i = 0; i < iters; i++) losses.push_back(MappingIteration(fr));
        return losses;
    }
    torch::NoGradGuard ng;
    c10::DeviceGuard guard(dev_);
    ensure_direct_(iters);
    Direct& d = *d_;
    const LoopFrame frame{fr.rgb.to(dev_, torch::kFloat32).contiguous(), fr.depth.to(dev_, torch::kFloat32).contiguous(), fr.Tcw};
    d.Tcw.copy_(fr.Tcw.to(torch::kFloat32).reshape({4, 4}));
    if (band_()) { // the frame's count of valid depth pixels: the one total the band's depth gradient divides by (a constant of the frame)
        d.frame_sums.zero_();
        d.frame_sums.slice(0, 2, 3).copy_((frame.depth > 0).sum().to(torch::kFloat32).reshape({1}));
    }
    if (shard_) { shard_preflight_(); d.order_stale = true; }
    else if (d.fresh) shard_preflight_(); // (unsharded too — ADVICE r4: inside a batch nobody looks; an iteration that overflowed is still retaken below,
                                          // but the iterations behind it ran with Adam step numbers one too high: size the workspace BEFORE the batch instead)
    d.fresh = false;
    // ONE iteration per call (MappingIteration in a caller's own loop): the loss is POSTED into host-mapped memory by the finish kernel — in the middle of the
    // iteration, before the backward — and the host spins on the word (the tracking loop's mechanism): no copy kernel, no stream synchronisation, the next call's
    // launches queue up behind this one's backward. Only a NaN (an overflowed forward posts one) takes the blocking look at the workspace. (round 6)
    if (iters == 1 && !shard_) {
        for (int attempt = 0; attempt < 4; attempt++) {
            __atomic_store_n(reinterpret_cast<uint32_t*>(d.posted), kNotPosted, __ATOMIC_RELEASE);
            direct_map_iteration_(frame, d.posted_dev);
            const double v = wait_posted(d.posted, (hipStream_t)stream_());
            if (!std::isnan(v) || !direct_overflowed_()) { losses.push_back(v); return losses; }
            for (int g = 0; g < 5; g++) fopt_->retract_step(g); // the step was skipped on the device: taken again on the grown workspace
        }
        throw std::runtime_error("direct loop: the binning workspace keeps overflowing");
    }
    int done = 0;
    while (done < iters) {
        const int batch = iters - done;
        for (int i = 0; i < batch; i++) direct_map_iteration_(frame, f(d.history) + i);
        const auto h = d.history.slice(0, 0, batch).to(torch::kCPU); // the one read-back (and synchronisation) of the batch
        // an iteration whose forward ran out of workspace rendered nothing, wrote NaN and skipped its step: it is taken again
        // (its step count with it) once the workspace has grown. The first iteration on a new map size is where this can happen.
        const bool overflowed = direct_overflowed_();
        if (shard_) { // the ranks step together or not at all: an overflow on ANY rank corrupted every rank's composite
            if (shard_any_(overflowed)) throw std::runtime_error("sharded loop: a rank's binning workspace overflowed inside a batch of iterations; raise LoopConfig::binning_capacity");
            for (int i = 0; i < batch; i++) losses.push_back(h[i].item<float>());
            break;
        }
        int kept = 0;
        for (int i = 0; i < batch; i++) {
            const double v = h[i].item<float>();
            if (overflowed && std::isnan(v)) continue;
            losses.push_back(v);
            kept++;
        }
        if (overflowed) for (int k = 0; k < batch - kept; k++) for (int g = 0; g < 5; g++) fopt_->retract_step(g);
        if (!overflowed) break;
        done += kept;
    }
    return losses;
}

std::vector<double> SlamLoop::direct_track_(const LoopFrame& fr, const torch::Tensor& Tcw_init, int iters, torch::Tensor* Tcw_best, const LoopMatches* matches)
{
    torch::NoGradGuard ng;
    c10::DeviceGuard guard(dev_);
    ensure_direct_(iters);
    Direct& d = *d_;
    void* const st = stream_();
    const LoopFrame frame{fr.rgb.to(dev_, torch::kFloat32).contiguous(), fr.depth.to(dev_, torch::kFloat32).contiguous(), fr.Tcw};
    // Gaussian::InitCameraPose (Gaussian.cc:97-150): quaternion of the initial rotation, its translation, fresh moments
    const auto T0 = Tcw_init.to(torch::kCPU, torch::kFloat32);
    const auto q0 = rot_to_quat(T0.slice(0, 0, 3).slice(1, 0, 3)).reshape({4});
    d.pose.copy_(torch::cat({q0, T0.slice(0, 0, 3).slice(1, 3, 4).reshape({3})}));
    d.pose_moments.zero_();
    d.best.zero_(); d.best.slice(0, 0, 1).fill_(std::numeric_limits<float>::infinity());
    chk(gsr_pose_from_quat(f(d.pose), f(d.pose) + 4, f(d.Tcw), st), "gsr_pose_from_quat");
    if (shard_ || d.fresh) shard_preflight_();
    d.fresh = false;
    // the map does not move while the pose is tracked: its activations are formed once per call
    if (d.n > 0) chk(gsr_map_prepare((size_t)d.n, nullptr, f(logit_opacities), f(log_scales), f(unnorm_quat), nullptr, nullptr, f(d.opac), f(d.scales), f(d.rots), 0.f, 0.f, 0.f,
                        nullptr, nullptr, st), "gsr_map_prepare");
    const float w3[3] = {(float)cfg_.im_weight_tracking, (float)cfg_.depth_weight_tracking, 0.f};
    const size_t HW = (size_t)H_ * W_;
    const float* const im0 = shard_ ? f(d.comp) : f(d.layers);
    const size_t gap = shard_ ? 4 : 0;
    const float *img = im0, *sil = im0 + 4 * HW + gap;
    const float* dep = cfg_.use_sur_depth ? nullptr : im0 + 3 * HW;
    const float* sur = cfg_.use_sur_depth ? im0 + 5 * HW + gap : nullptr;
    // the feature front end's matches: the reprojection term is added to the loss and to one pose row by one launch per iteration
    torch::Tensor m_obs, m_X, m_s2, m_inl;
    float m_cx = 0.f, m_cy = 0.f;
    if (matches && matches->obs.defined() && matches->obs.size(0) > 0) {
        m_obs = matches->obs.to(dev_, torch::kFloat32).reshape({-1, 2}).contiguous(); m_X = matches->Xw.to(dev_, torch::kFloat32).reshape({-1, 3}).contiguous();
        m_s2 = matches->inv_sigma2.to(dev_, torch::kFloat32).reshape({-1}).contiguous();
        if (m_X.size(0) != m_obs.size(0) || m_s2.size(0) != m_obs.size(0)) throw std::runtime_error("Track: matches of different lengths");
        m_inl = torch::ones({m_obs.size(0)}, torch::TensorOptions().device(dev_).dtype(torch::kUInt8));
        m_cx = (float)(matches->cx >= 0 ? matches->cx : (W_ - 1) / 2.0); m_cy = (float)(matches->cy >= 0 ? matches->cy : (H_ - 1) / 2.0);
    }
    const int feature_clear = (int)(iters / 2.0);                                                       // Render.cc:1051
    // (a sharded run sums the ranks' pose rows: a term every rank holds in full enters each rank's row with weight 1 / world)
    // (the band exchange sums the ranks' loss words with the pose rows: the term's VALUE is added on rank 0 only, the other ranks add it to a word nobody reads)
    const bool band = band_();
    // (round 6) unsharded, on the surface depth: the loss needs the colours, the median depth and a silhouette MASK — the plain forward renders the first two and
    // keeps the final transmittance per pixel (1 - T is the silhouette): no fused depth / silhouette channels, neither forwards nor backwards
    // (sharded, on the surface depth: the compositor needs every layer's colours, silhouette and surface depth — the plain forward with gsr_forward_args.out_sil
    // renders exactly those; the blended-depth plane of the layer is not written and not exchanged, zeroed once so that the compositor's unused depth channel stays finite)
    const bool shard_plain = shard_ && band_() && cfg_.use_sur_depth && cfg_.fused_update && !std::getenv("GSR_EXP_TRACK_DUAL_BWD");
    if (shard_plain) d.layers.slice(0, 3, 4).zero_();
    const bool plain_track = !shard_ && cfg_.use_sur_depth && d.n > 0 && !std::getenv("GSR_EXP_TRACK_DUAL_BWD"); // (an empty map renders nothing: no transmittance plane either — the zeroed layers then mask every pixel out)
    float* final_T = nullptr;
    if (plain_track) chk(gsr_transmittance_view(b(d.image), W_, H_, &final_T), "gsr_transmittance_view");
    float* const band_sums = f(d.pose_acc) + 64 * 12 + 8; // [8] this rank's tracking-loss sums: words 776..783 of the all-reduced block
    float* const loss_word = band ? (d.rank == 0 ? band_sums + 5 : f(d.frame_sums) + 7) : f(d.sums) + 5;
    auto reproj = [&](int it, float* row) {
        if (!m_obs.defined()) return;
        chk(gsr_reproj_loss(f(m_obs), f(m_X), f(m_s2), (size_t)m_obs.size(0), f(d.Tcw), fx_, fy_, m_cx, m_cy, (float)cfg_.feature_weight_tracking,
                            shard_ ? 1.f / (float)d.world : 1.f, it < feature_clear ? 2 : it == feature_clear ? 1 : 0, m_inl.data_ptr<uint8_t>(), row,
                            loss_word, st), "gsr_reproj_loss");
    };
    std::vector<double> history;
    double last_loss = 0.0;
    int step = 0;
    for (int it = 0; it < iters; it++) {
        direct_forward_(true, false, 0.f, plain_track || shard_plain, shard_plain); // (the camera transform of Render.cc:750-752 rides in the projection kernel)
        if (band) band_forward_(true, kBandTrack);
        else if (shard_) shard_composite_forward_(true, false);
        // Render.cc:1088-1105: the masked L1 sums and their gradient planes, one pass over the render (band exchange: over this rank's band of rows; its
        // sums are added up over the ranks by the pose rows' all-reduce)
        if (band)
            chk(gsr_track_loss_rows(img, dep, sur, sil, f(frame.rgb), f(frame.depth), H_, W_, 0.99f, w3, f(d.loss_partial), band_sums, f(d.g_image), f(d.g_ds),
                                    reinterpret_cast<uint32_t*>(d.tickets.data_ptr<int>()), d.b0, d.b1, 0, st), "gsr_track_loss_rows");
        else if (plain_track)
            chk(gsr_track_loss_rows(img, dep, sur, final_T, f(frame.rgb), f(frame.depth), H_, W_, 0.99f, w3, f(d.loss_partial), f(d.sums), f(d.g_image), f(d.g_ds),
                                    reinterpret_cast<uint32_t*>(d.tickets.data_ptr<int>()), 0, H_, 1, st), "gsr_track_loss_rows (transmittance)");
        else
            chk(gsr_track_loss(img, dep, sur, sil, f(frame.rgb), f(frame.depth), H_, W_, 0.99f, w3, f(d.loss_partial), f(d.sums), f(d.g_image), f(d.g_ds),
                               reinterpret_cast<uint32_t*>(d.tickets.data_ptr<int>()), st), "gsr_track_loss");
        gsr_pose_update_args u{};
        u.quat_trans = f(d.pose); u.moments = f(d.pose_moments); u.best = f(d.best); u.history = d.posted_dev + it; u.Tcw = f(d.Tcw);
        u.partial = f(d.pose_partial); u.loss = f(d.sums) + 5; u.geom = b(d.geom);
        __atomic_store_n(reinterpret_cast<uint32_t*>(d.posted + it), kNotPosted, __ATOMIC_RELEASE);
        u.lr = cfg_.lr_cam_quat; u.beta1 = 0.9; u.beta2 = 0.999; u.eps = 1e-15; u.step = ++step;
        uint32_t* const tickets = reinterpret_cast<uint32_t*>(d.tickets.data_ptr<int>()) + GSR_TICKET_WORDS;
        if (shard_ && cfg_.fused_update) { // the shard's pose sums come out of the backward's per-splat stage (accumulator rows), are summed over the ranks, and a one-wave kernel takes the (replicated) step
            if (band) { band_backward_(kBandTrack); u.loss = band_sums + 5; } // (the loss the step records is the all-reduced word behind the rows)
            else shard_composite_backward_();
            u.partial = f(d.pose_acc);
            //  whether this iteration counts is decided by ALL ranks: the per-splat stage leaves the rank's overflow flag behind the rows, the
            // all-reduce sums it with them, and every rank's step kernel skips + posts NaN on the total — so every rank then enters shard_any_ below
            u.skip = f(d.pose_acc) + 64 * 12;
            reproj(it, f(d.pose_acc)); // (the rows are zero here; the term enters every rank's row with weight 1 / world)
            gsr_pose_step_args ps{f(xyz), &u, 1, u.skip};
            direct_backward_(true, true, nullptr, &ps);
            d_->all_reduce(d.pose_acc);
            chk(gsr_pose_finish(&u, f(d.pose_acc), f(d.last_sums), st), "gsr_pose_finish");
        } else if (shard_) { // every rank holds the pose sums of its shard: the rows are summed over the ranks before the (replicated) pose step
            shard_composite_backward_();
            direct_backward_(true, true, nullptr, nullptr);
            if (d.n > 0) chk(gsr_pose_grad(f(xyz), f(d.d_mc), (size_t)d.n, f(d.Tcw), f(d.pose_partial), nullptr, st), "gsr_pose_grad");
            else d.pose_partial.zero_();
            reproj(it, f(d.pose_partial));
            { // the rank's overflow flag (word 1 of the geometry header) behind the rows: summed over the ranks with them, read back as the step's skip flag
                auto flag = d.pose_partial_buf.slice(0, GSR_POSE_PARTIALS * 12, GSR_POSE_PARTIALS * 12 + 1);
                if (d.n > 0) flag.copy_(d.geom.slice(0, 0, 8).view(torch::kInt32).slice(0, 1, 2)); else flag.zero_();
                u.skip = f(flag);
            }
            d_->all_reduce(d.pose_partial_buf);
            chk(gsr_pose_update(&u, st), "gsr_pose_update");
        } else if (cfg_.fused_update) { // the backward's per-splat stage forms the pose sums (into accumulator rows that are zero between launches), a one-wave kernel takes the step: no dL/dmeans tensor
            u.partial = f(d.pose_acc);
            reproj(it, f(d.pose_acc)); // (the accumulator rows are zero here: the per-splat stage adds to them, the one-wave kernel behind it sums them)
            const gsr_pose_step_args ps{f(xyz), &u, 0, nullptr};
            direct_backward_(true, true, nullptr, &ps); // the [z, 1, 0] colours are detached while tracking (Render.cc:949-981)
        } else {
            direct_backward_(true, true, nullptr, nullptr);
            if (m_obs.defined()) { // (the term goes between the rows and the step: two launches instead of gsr_pose_step's one)
                chk(gsr_pose_grad(f(xyz), f(d.d_mc), (size_t)d.n, f(d.Tcw), f(d.pose_partial), nullptr, st), "gsr_pose_grad");
                reproj(it, f(d.pose_partial));
                chk(gsr_pose_update(&u, st), "gsr_pose_update");
            } else
                chk(gsr_pose_step(f(xyz), f(d.d_mc), (size_t)d.n, &u, tickets, st), "gsr_pose_step"); // the pose sums and the step in one launch
        }
        const double lv = wait_posted(d.posted + it, (hipStream_t)st); // Render.cc:1107: the loop looks at every loss
        if (shard_) {
            if (std::isnan(lv) && shard_any_(direct_overflowed_()))
                throw std::runtime_error("sharded loop: a rank's binning workspace overflowed while tracking; raise LoopConfig::binning_capacity");
        } else if (std::isnan(lv) && direct_overflowed_()) { --step; --it; continue; } // the workspace has grown: take the iteration again
        history.push_back(lv);
        if (std::fabs(last_loss - lv) < 10e-4) break;                                                    // Render.cc:1113-1114
        last_loss = lv;
    }
    if (Tcw_best) {
        const auto bh = d.best.to(torch::kCPU);
        *Tcw_best = rt2T(bh.slice(0, 1, 5).reshape({4, 1}), bh.slice(0, 5, 8).reshape({3, 1})).to(dev_);
    }
    return history;
}

int SlamLoop::shard_rank_() const { return d_ ? d_->rank : 0; }

torch::Tensor SlamLoop::LastPoseSums() const
{
    if (!d_ || !d_->pose_partial.defined()) throw std::runtime_error("LastPoseSums: no tracking iteration yet");
    if (shard_ && cfg_.fused_update) return d_->last_sums.clone(); // (the sharded loop's step kernel leaves the twelve sums it used there)
    return d_->pose_partial.sum(0);
}

// Sharded: does ANY rank say yes? (a host-side decision every rank must take alike)
bool SlamLoop::shard_any_(bool mine)
{
    Direct& d = *d_;
    if (!d.pg) return mine;
    auto t = torch::full({1}, mine ? 1.f : 0.f, torch::TensorOptions().dtype(torch::kFloat32).device(d.staged ? torch::Device(torch::kCPU) : dev_));
    std::vector<torch::Tensor> v{t};
    d.pg->allreduce(v)->wait();
    return t.item<float>() > 0.f;
}

// Sharded: one forward of the shard under the current pose before the first iteration on a workspace, so that every rank's binning workspace holds its
// frame with room to spare (inside a batch nobody looks, and an overflow on one rank would spoil every rank's composite).
void SlamLoop::shard_preflight_()
{
    Direct& d = *d_;
    for (int attempt = 0; attempt < 3; attempt++) {
        if (d.n > 0) {
            chk(gsr_map_prepare((size_t)d.n, f(xyz), f(logit_opacities), f(log_scales), f(unnorm_quat), f(d.Tcw), f(d.mc), f(d.opac), f(d.scales), f(d.rots), 0.f, 0.f, 0.f,
                                nullptr, nullptr, stream_()), "gsr_map_prepare");
            direct_forward_();
        }
        int R = 0, ov = 0;
        if (d.n > 0) chk(gsr_ws_status(b(d.geom), stream_(), &R, &ov), "gsr_ws_status");
        const size_t want = 2 * (size_t)R + 65536;
        const bool grow = d.n > 0 && gsr_binning_bytes(want) > d.binning_bytes && (ov || gsr_binning_bytes((size_t)R + (size_t)R / 2) > d.binning_bytes);
        if (grow) grow_binning_(want);
        if (!shard_any_(ov != 0)) return;
    }
    throw std::runtime_error("sharded loop: the binning workspace keeps overflowing");
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> SlamLoop::RenderComposite(const torch::Tensor& Tcw)
{
    if (!direct_()) throw std::runtime_error("RenderComposite needs LoopConfig::direct");
    torch::NoGradGuard ng;
    c10::DeviceGuard guard(dev_);
    ensure_direct_(1);
    Direct& d = *d_;
    d.Tcw.copy_(Tcw.to(torch::kFloat32).reshape({4, 4}));
    if (shard_) shard_preflight_();
    if (d.n > 0)
        chk(gsr_map_prepare((size_t)d.n, f(xyz), f(logit_opacities), f(log_scales), f(unnorm_quat), f(d.Tcw), f(d.mc), f(d.opac), f(d.scales), f(d.rots), 0.f, 0.f, 0.f,
                            nullptr, nullptr, stream_()), "gsr_map_prepare");
    direct_forward_();
    if (!shard_) return {d.layers.slice(0, 0, 3).clone(), d.layers.slice(0, 5, 6).clone(), d.layers.slice(0, 3, 5).clone()};
    shard_composite_forward_(true, false);
    const int64_t HW = (int64_t)H_ * W_;
    const auto plane = [&](int64_t off, int64_t n) { return d.comp.slice(0, off, off + n * HW).reshape({n, H_, W_}); };
    return {plane(0, 3).clone(), plane(5 * HW + 4, 1).clone(), torch::cat({plane(3 * HW, 1), plane(4 * HW + 4, 1)}, 0)};
}

torch::Tensor SlamLoop::ShardRenderStep(const torch::Tensor& Tcw, const torch::Tensor& G, int preflight)
{
    if (!direct_() || !shard_) throw std::runtime_error("ShardRenderStep needs a sharded direct loop (SetShard)");
    torch::NoGradGuard ng;
    c10::DeviceGuard guard(dev_);
    ensure_direct_(1);
    Direct& d = *d_;
    void* const st = stream_();
    if (!G.is_cuda() || G.scalar_type() != torch::kFloat32 || !G.is_contiguous() || G.numel() != (int64_t)5 * H_ * W_) throw std::runtime_error("ShardRenderStep: G must be a contiguous float32 device tensor [5,H,W]");
    //  the pose is copied on EVERY call (64 bytes, no synchronisation: a caller may update its tensor in place), and the pre-flight — it contains
    // a collective — follows a condition every rank evaluates alike: the caller's argument, or the first call on a (re)built workspace
    d.Tcw.copy_(Tcw.to(torch::kFloat32).reshape({4, 4}));
    if (preflight > 0 || (preflight < 0 && d.fresh)) shard_preflight_();
    d.fresh = false;
    if (d.n > 0)
        chk(gsr_map_prepare((size_t)d.n, f(xyz), f(logit_opacities), f(log_scales), f(unnorm_quat), f(d.Tcw), f(d.mc), f(d.opac), f(d.scales), f(d.rots), 0.f, 0.f, 0.f,
                            nullptr, nullptr, st), "gsr_map_prepare");
    direct_forward_();
    if (band_()) { // the band exchange: the caller's gradient of the composite taken back on this rank's band of rows, for every rank's layer
        band_forward_(true, kBandRender);
        const torch::Tensor keep = d.G;
        d.G = G;
        band_backward_(kBandRender, f(G) + (size_t)4 * H_ * W_);
        d.G = keep;
        direct_backward_(false, false, nullptr, nullptr);
        if (d.n > 0) chk(gsr_pose_grad(f(xyz), f(d.d_mc), (size_t)d.n, f(d.Tcw), f(d.pose_partial), nullptr, st), "gsr_pose_grad");
        else d.pose_partial.zero_();
        d.all_reduce(d.pose_partial);
        return d.pose_partial;
    }
    shard_composite_forward_(true, false);
    // the compositor's backward reads the loss's gradient from d.G: hand it the caller's (the silhouette's upstream gradient rides in plane 4)
    const torch::Tensor keepG = d.G;
    d.G = G;
    const long long* const order = (const long long*)d.order.data_ptr<int64_t>();
    d.stream = (hipStream_t)st;
    chk(gsr_composite_backward_local(d.world, d.rank, order, f(d.gathered), 2, f(d.layers), f(d.G), H_, W_, f(d.D), f(d.c_own), st), "gsr_composite_backward_local");
    d.all_gather(d.c_all, d.c_own);
    chk(gsr_composite_backward_occlusion(d.world, d.rank, order, f(d.gathered), 2, f(d.c_all), f(d.G) + (size_t)4 * H_ * W_, H_, W_, f(d.D) + (size_t)4 * H_ * W_, st),
        "gsr_composite_backward_occlusion");
    d.G = keepG;
    direct_backward_(false, false, nullptr, nullptr);
    if (d.n > 0) chk(gsr_pose_grad(f(xyz), f(d.d_mc), (size_t)d.n, f(d.Tcw), f(d.pose_partial), nullptr, st), "gsr_pose_grad");
    else d.pose_partial.zero_();
    d.all_reduce(d.pose_partial);
    return d.pose_partial;
}

// Sharded: the cell of the k-d partition every point lies in ([n] int64 ranks) — the owner rule of map growth
torch::Tensor SlamLoop::shard_cells_(const torch::Tensor& pts) const
{
    const Direct& d = *d_;
    const int64_t n = pts.size(0);
    if (d.world == 1) return torch::zeros({n}, torch::TensorOptions().device(pts.device()).dtype(torch::kInt64));
    if (!d.kd_nodes.defined()) throw std::runtime_error("map growth on a sharded loop needs the k-d partition (SetShard's kd_nodes)");
    const auto kd = d.kd_nodes.to(torch::kCPU);
    auto node = torch::zeros({n}, torch::TensorOptions().device(pts.device()).dtype(torch::kInt64));
    // (the nodes are numbered parents before children: one pass over them settles every point)
    for (int64_t i = 0; i < kd.size(0); i++) {
        const int axis = (int)kd[i][0].item<float>();
        const float split = kd[i][1].item<float>();
        const int64_t left = (int64_t)kd[i][2].item<float>(), right = (int64_t)kd[i][3].item<float>();
        const auto here = node == i, go_left = pts.select(1, axis) < split;
        node = torch::where(here & go_left, torch::full_like(node, left), torch::where(here & ~go_left, torch::full_like(node, right), node));
    }
    return -1 - node;
}

} // namespace ORB_SLAM2
