// SlamLoop.h — the tracking and mapping loops of GSORB-SLAM's Render / Gaussian classes as a libtorch (C++) driver on top of
// the drop-in operator (Rasterizer.h): host code in C++, like the reference's (src/Render.cc:402-493 RenderForFrame /
// mapping iterations, :985-1141 RenderStartTraking; src/Gaussian.cc:144-175 optimisers; include/Utils.h:56-77 rt2T;
// src/Utils.cc:39-100 losses). What a maintainer would keep of Render.cc once the rasterizer underneath is this library: the
// loop bodies, with the two renders of an iteration fused into GaussianRasterizer::forward_pair.
// The Python twin (same arithmetic, used by the parity tests) is gsorb-slam_amd/harness.py.
#pragma once

#include <torch/torch.h>

#include <memory>
#include <vector>

#include "FusedOps.h"
#include "Rasterizer.h"

namespace c10d { class ProcessGroup; }
struct gsr_map_update_args; // include/gsr.h
struct gsr_pose_step_args;  // include/gsr.h

namespace ORB_SLAM2 {

// Examples/RGB-D/replica.yaml:87-117 (Mapping / Tracking blocks)
struct LoopConfig {
    double im_weight_mapping = 1.0, depth_weight_mapping = 0.7, sur_depth_weight_mapping = 0.35, reg_long_weight = 5.0,
           reg_scalar_weight = 10.0, lam = 0.8;
    double lr_mean3d = 0.0001, lr_rgb = 0.0025, lr_rotation = 0.001, lr_opacities = 0.05, lr_scales = 0.001;
    double lr_cam_quat = 0.0004; // used for BOTH pose groups, like the reference (Gaussian.cc:149-150)
    double im_weight_tracking = 0.7, depth_weight_tracking = 1.0;
    double feature_weight_tracking = 0.1; // Tracking.FeatureWeight (replica.yaml, scannet.yaml): weight of the ORB matches' reprojection term (Render.cc:1096)
    double scale_modifier = 1.0, scene_radius = 1.0;
    double prune_opacities = 0.005, median_mul = 40.0; // Mapping.PruneOpcities / Mapping.MedianMul
    int init_scalar_method = 2;                        // 0 Distance, 1 DistanceMean, 2 SinglePixel (Gaussian.cc:59-79)
    bool use_sur_depth = true;
    bool fused_pair = true; // one rasterizer pass per iteration (forward_pair); false: two passes like Render.cc:927-981
    bool fused_ops = true;  // camera transform, pose matrix, SSIM and Adam through the loop kernels of the C ABI (FusedOps.h);
                            // false: the reference's plain libtorch arithmetic (matmul, scalar-tensor rt2T, conv2d, torch::optim::Adam)
    bool direct = true;     // (with fused_pair and fused_ops) an iteration is a fixed sequence of C-ABI launches on a persistent workspace
                            // — no autograd graph, no allocation, no gradient tensors of the raw parameters (DirectLoop.cpp);
                            // false: the same kernels through libtorch autograd
    int64_t binning_capacity = 0; // (direct) tile instances the rasterizer's binning workspace starts with; 0 = 4 x the map's size + 65536. A forward
                                  // that needs more skips its iteration's step on the device, the host grows the workspace at its next read and takes
                                  // the iteration again: any value gives the same results
    bool fused_loss = true;   // (direct) the mapping loss as gsr_map_loss_forward / _finish / _backward (SSIM and the pixel terms in the same two passes);
                              // false: gsr_pixel_loss, gsr_ssim_*, gsr_pixel_loss_backward_add, gsr_map_loss_total as six launches
    bool fused_update = true; // (direct) the backward's per-splat stage takes the Adam step itself (gsr_backward_args.fused_map_update);
                              // false: gsr_backward writes the gradients, gsr_map_update reads them
    bool band_exchange = true; // (sharded, with fused_loss and fused_update) the BAND exchange of round 6: every rank receives all ranks' layers for its band of pixel rows
                               // (one grouped point-to-point exchange), composites, evaluates the loss and differentiates the composite there — for every rank's layer —
                               // and returns each rank its rows (a second exchange): all per-pixel work / world, two collectives per mapping iteration instead of three,
                               // a third of the bytes at 8 ranks. false: round 5's replicated composite (all-gather, all-reduce, all-gather; every rank the whole frame)
};

struct LoopFrame {
    torch::Tensor rgb;   // [3,H,W]
    torch::Tensor depth; // [H,W], 0 = invalid
    torch::Tensor Tcw;   // [4,4]
};
// The feature front end's matches for a tracking call (Render.cc:1005-1043: map points matched to the frame's keypoints)
struct LoopMatches {
    torch::Tensor obs;        // [M,2] observed pixels (u, v)
    torch::Tensor Xw;         // [M,3] world points
    torch::Tensor inv_sigma2; // [M] inverse level variance of the keypoints
    double cx = -1, cy = -1;  // principal point (< 0: the image centre the loop's other back-projections use, (W - 1) / 2, (H - 1) / 2)
};

class SlamLoop {
public:
    SlamLoop(const LoopConfig& cfg, int width, int height, float fx, float fy, torch::Device device);
    ~SlamLoop();

    // Gaussian::GaussianOptimizer (Gaussian.cc:152-175): five Adam groups, eps 1e-15
    void SetMap(torch::Tensor xyz, torch::Tensor rgb, torch::Tensor unnorm_quat, torch::Tensor logit_opacities, torch::Tensor log_scales);

    // Render.cc:1054-1126: pose-only optimisation against one frame; returns the loss of every iteration that ran and the
    // best pose (lowest loss) in *Tcw_best
    // With `matches` the objective is the reference's whole tracking loss: + feature_weight_tracking * Lrpj, the reprojection error of the matches
    // under the pose being optimised, outliers frozen out halfway through (Render.cc:1031-1096; gsr_reproj_loss).
    std::vector<double> Track(const LoopFrame& frame, const torch::Tensor& Tcw_init, int iters, torch::Tensor* Tcw_best, const LoopMatches* matches = nullptr);

    // Render.cc:420-483: one mapping iteration on one keyframe (loss, backward, Adam step); returns the loss
    double MappingIteration(const LoopFrame& frame);
    // The loop of Render::RenderForFrame (Render.cc:418-483) on one keyframe: `iters` mapping iterations. Like the reference's loop
    // it never looks at a loss in between: the losses of all iterations are read back once, at the end.
    std::vector<double> MapFrame(const LoopFrame& frame, int iters);

    // ---- map growth (the other half of the per-frame loop: Render.cc:557-616, Gaussian.cc:40-95, :180-258) ----
    // Render::AddGaussian + ProjectPixel + Gaussian::AddGaussianPoints: renders the frame's view, masks the pixels the map does
    // not explain (silhouette < 0.8, or not certain AND dark AND depth error above avg + MedianMul * median), back-projects
    // them through the frame's depth and appends them as new Gaussians (logit opacity 1, identity quaternion, the configured
    // scale init), Adam moments extended with zeros. Returns the number of Gaussians added.
    int64_t AddGaussians(const LoopFrame& frame);
    // Gaussian::AddGaussianPoints on given world points / colours ([n,3] each)
    void AddPoints(const torch::Tensor& pts, const torch::Tensor& cols);
    // Render::RemoveGaussian: drops the Gaussians whose opacity fell below prune_opacities, with their Adam moments.
    int64_t PruneLowOpacity();
    int64_t size() const { return xyz.defined() ? xyz.size(0) : 0; }

    // ---- the map sharded over the GPUs of one node (multi-GPU scheme B; DirectLoop.cpp) ----
    // This loop then holds ONE SHARD of the map (SetMap with the shard's rows): every iteration rasterizes the shard and composites the ranks'
    // layers front to back around three collectives of `group` (backend "nccl" = RCCL; any other backend is staged through the host); the loss is
    // evaluated on the composite on every rank, the regulariser sums (mapping) and the pose rows (tracking) are all-reduced, Adam state stays
    // local. kd_nodes [world - 1, 4] = the k-d partition whose cell `rank` this shard is (gsr_shard_order; sharded.KdPartition.nodes); undefined:
    // the ranks are already in front-to-back order (depth slabs of one view). group = null with world 1: the same launch sequence, no exchange.
    // The reference is single-GPU (src/Render.cc:402-483, :1054-1126): north_star's "shard Gaussians, all-reduce pose / loss gradients only".
    void SetShard(c10::intrusive_ptr<c10d::ProcessGroup> group, int rank, int world, const torch::Tensor& kd_nodes);
    // the whole map's render for a pose from every rank's shard: {colour [3,H,W], surface depth [1,H,W], depth / silhouette [2,H,W]} (collective call)
    std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> RenderComposite(const torch::Tensor& Tcw);
    // The sharded rasterize path forwards and backwards in one call (collective; what bench.py's N > 1 headline times): the own cell's render, the
    // composite of all ranks' layers, an upstream gradient of the composite G [5,H,W] = d/d(rgb, depth, silhouette) taken back through the compositor and
    // the rasterizer to this rank's Gaussians (gradient buffers of the workspace) and to the pose — returns the [GSR_POSE_PARTIALS, 12] pose rows summed over the
    // ranks (their column sums are dL/dR row-major, dL/dt). Launches only: nothing here synchronises with the host.
    // preflight: 1 = size every rank's binning workspace for this pose first (one forward + a collective), 0 = never, -1 = on the first call after SetShard or a
    // change of the map's size. Every rank must pass the same value (the pre-flight contains a collective).
    torch::Tensor ShardRenderStep(const torch::Tensor& Tcw, const torch::Tensor& G, int preflight = -1);
    // Re-balance of a sharded map (sharded.rebalance_loop drives the exchange): every Gaussian of this loop with what travels with it — raw
    // parameters (14 floats: xyz, rgb, quaternion, logit, log-scales), exp_avg (14), exp_avg_sq (14): [n, 42] —, and the surgery that keeps the
    // rows `keep` and appends the rows that arrive from other ranks WITH their moments (Gaussian.cc:218-258 does the same with zeros / a selection)
    torch::Tensor ExportRows();
    void ReplaceRows(const torch::Tensor& keep, const torch::Tensor& arrivals);
    // (inspection) the twelve pose sums dL/dR (row-major), dL/dt of the last tracking iteration that went through gsr_pose_grad's rows — the
    // sharded loop (summed over the ranks) and the unsharded one with LoopConfig::fused_update = false
    torch::Tensor LastPoseSums() const;
    std::string ShardTransport() const;

    // both renders of an iteration: {colour [3,H,W], surface (median) depth [1,H,W], depth/silhouette [2,H,W]}
    std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> RenderPair(const torch::Tensor& Tcw, bool tracking);

    torch::Tensor xyz, rgb, unnorm_quat, logit_opacities, log_scales;

private:
    LoopConfig cfg_;
    int W_, H_;
    float fx_, fy_;
    torch::Device dev_;
    GaussianRasterizer rasterizer_;
    std::unique_ptr<torch::optim::Adam> opt_, opt_pose_;
    std::unique_ptr<fused::Adam> fopt_, fopt_pose_;
    torch::Tensor cam_quat_, cam_trans_, taps_;
    std::vector<float> taps_host_;
    std::vector<torch::Tensor> act_; // Track(): the map's activations (opacity, scales, unit quaternions), formed once per call
    std::vector<torch::Tensor*> params_();
    // LoopConfig::direct (DirectLoop.cpp)
    struct Direct;
    std::shared_ptr<Direct> d_; // (shared_ptr: its deleter is bound where the type is complete)
    bool direct_() const { return cfg_.direct && cfg_.fused_pair && cfg_.fused_ops; }
    bool shard_ = false;
    void shard_composite_forward_(bool pose_moved, bool with_reg);
    void shard_composite_backward_();
    bool band_() const;
    enum { kBandMap = 0, kBandTrack = 1, kBandRender = 2 }; // which exchange: a mapping iteration (ten halo rows), a tracking iteration (no blended depth on the surface depth), ShardRenderStep (every plane)
    void band_forward_(bool pose_moved, int kind);
    void band_backward_(int kind, const float* g_sil = nullptr);
    void shard_preflight_();
    bool shard_any_(bool mine);
    torch::Tensor shard_cells_(const torch::Tensor& pts) const;
    int shard_rank_() const;
    void* stream_() const;
    void ensure_direct_(int64_t history_len);
    void grow_binning_(size_t capacity);
    void direct_forward_(bool from_world = false, bool raw = false, float reg_limit = 0.f, bool plain = false, bool plain_sil = false);
    void direct_backward_(bool detach_depth_colour, bool means_only, const ::gsr_map_update_args* fused, const ::gsr_pose_step_args* pose_step);
    bool direct_overflowed_();
    void direct_map_iteration_(const LoopFrame& frame, float* loss_slot);
    std::vector<double> direct_track_(const LoopFrame& frame, const torch::Tensor& Tcw_init, int iters, torch::Tensor* Tcw_best, const LoopMatches* matches);
    void replace_params_(const std::vector<torch::Tensor>& fresh, int64_t added, const torch::Tensor* keep);
};

// include/Utils.h:56-77: Tcw [4,4] from an un-normalised quaternion (r,x,y,z) [4,1] and a translation [3,1]
torch::Tensor rt2T(const torch::Tensor& quat, const torch::Tensor& trans);
// cv::Quatd::createFromRotMat as Gaussian::InitCameraPose uses it (Gaussian.cc:97-128)
torch::Tensor rot_to_quat(const torch::Tensor& R);

} // namespace ORB_SLAM2
