"""Generates tests/golden/ref_eval.npz by IMPORTING the reference's own runnable Python:

  scripts/eval_ate.py            align (:6-33), read_trajectory_file (:35-54), evaluate_ate (:56-82)
  Thirdparty/diff_gaussian_rasterization/utils/sh_utils.py   RGB2SH / SH2RGB (:114-118)

Run only in the build container (needs /root/reference); the .npz is committed. The fixture holds inputs
made here (random trajectories, the text of a trajectory file written by THIS script) and the outputs the
reference functions returned for them — no reference source.
"""
import contextlib
import importlib.util
import io
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _rot(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def main():
    ate = _load("/root/reference/scripts/eval_ate.py", "ref_eval_ate")
    sh = _load("/root/reference/Thirdparty/diff_gaussian_rasterization/utils/sh_utils.py", "ref_sh_utils")
    rng = np.random.default_rng(77)
    out = {}
    # --- trajectories: a smooth path, an estimate = rigidly moved + noisy copy, one inf pose, unequal lengths
    N = 40
    t = np.linspace(0, 1, N)
    pos = np.stack([np.sin(2 * t), 0.3 * t, np.cos(3 * t) - 1], 1)
    gt = np.tile(np.eye(4), (N, 1, 1))
    gt[:, :3, 3] = pos
    for i in range(N):
        gt[i, :3, :3] = _rot(rng)
    Rg, tg = _rot(rng), np.array([0.4, -0.2, 1.0])
    est = np.tile(np.eye(4), (N - 3, 1, 1))
    est[:, :3, 3] = (pos[:N - 3] @ Rg.T + tg) + 0.01 * rng.standard_normal((N - 3, 3))
    est[:, :3, :3] = gt[:N - 3, :3, :3]
    est[5, 0, 3] = np.inf
    out["gt"], out["est"] = gt, est
    with contextlib.redirect_stdout(io.StringIO()):
        out["ate"] = np.float64(ate.evaluate_ate(gt, est))
    ok = [i for i in range(N - 3) if i != 5]
    R, tr, err = ate.align(gt[ok, :3, 3].T, est[ok, :3, 3].T)
    out["align_R"], out["align_t"], out["align_err"] = R, tr, err
    # a reflection-prone (planar, mirrored) case exercises the det < 0 branch
    m = rng.standard_normal((3, 12)); m[2] = 0
    d = m.copy(); d[0] = -d[0]
    R2, t2, e2 = ate.align(m, d)
    out["refl_model"], out["refl_data"], out["refl_R"], out["refl_t"], out["refl_err"] = m, d, R2, t2, e2
    # --- trajectory file: 16 numbers, 17 numbers (timestamp), comment, short line
    lines = ["# comment line"]
    for i in range(6):
        v = " ".join("%.9g" % x for x in gt[i].ravel())
        lines.append(("%d.5 " % i + v) if i % 2 else v)
    lines.append("1 2 3")
    text = "\n".join(lines) + "\n"
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(text)
    out["traj_text"] = np.array(text)
    out["traj_parsed"] = ate.read_trajectory_file(f.name)
    os.unlink(f.name)
    # --- SH <-> RGB
    rgb = rng.uniform(0, 1, (32, 3))
    out["rgb"], out["rgb2sh"], out["sh2rgb"] = rgb, sh.RGB2SH(rgb), sh.SH2RGB(sh.RGB2SH(rgb) * 0.7)
    np.savez_compressed(os.path.join(HERE, "ref_eval.npz"), **out)
    print("wrote ref_eval.npz", {k: np.asarray(v).shape for k, v in out.items()}, "ate", out["ate"])


if __name__ == "__main__":
    main()
