#!/usr/bin/env python3
"""Run the sliding-window backend over a stream and write the TUM pose log the reference writes (System.cpp:438).

  python tools/run_stream.py --sim-dir <dir>      # reference simulator output: imu_pose.txt + keyframe/all_points_<n>.txt
  python tools/run_stream.py --frames 60          # synthetic stream on the simulator's trajectory
  options: --lib hip|oracle|ref  --triangulate  --nonkey-every N  --out pose_output.txt  --export-sim-dir <dir>

Prints the APE the reference evaluates with (`evo_ape tum ground-truth.txt pose_output.txt -va`) when ground truth is
available.  The front end (feature tracking, initialisation) is out of scope: tracks come from the files' feature ids,
initial poses from the ground-truth columns perturbed by --pos-noise / --rot-noise."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ORACLE_DIR, load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sim-dir")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--landmarks-per-frame", type=int, default=30)
    ap.add_argument("--lib", choices=("hip", "oracle", "ref"), default="hip",
                    help="hip: the product; oracle: the CPU restatement; ref: the reference's own backend compiled into oracle/_ref (test infrastructure)")
    ap.add_argument("--triangulate", action="store_true")
    ap.add_argument("--nonkey-every", type=int, default=0)
    ap.add_argument("--pos-noise", type=float, default=0.02)
    ap.add_argument("--rot-noise", type=float, default=0.005)
    ap.add_argument("--out", default="pose_output.txt")
    ap.add_argument("--export-sim-dir")
    a = ap.parse_args()
    vio = load_package()
    if a.sim_dir:
        st = vio.stream.SimulatorFileStream(a.sim_dir)
        if not st.has_ground_truth:
            sys.exit("the IMU file has no ground-truth pose columns: no initial guesses (initialisation is out of scope)")
    else:
        st = vio.stream.SyntheticStream(n_frames=a.frames, landmarks_per_frame=a.landmarks_per_frame)
    if a.export_sim_dir:
        vio.stream.write_simulator_files(st, a.export_sim_dir)
    if a.lib == "hip":
        lib = vio.load_hip()
    elif a.lib == "ref":
        lib = vio.VioLib(os.path.join(ORACLE_DIR, "_ref", "libvio_ref.so"), "vior_")
    else:
        lib = vio.VioLib(os.path.join(ORACLE_DIR, "liboracle.so"), "vioo_")
    drv = vio.stream.StreamDriver(lib, st, pos_noise=a.pos_noise, rot_noise=a.rot_noise, triangulate=a.triangulate,
                                  nonkey_every=a.nonkey_every)
    t = time.perf_counter()
    traj = drv.run()
    dt = time.perf_counter() - t
    vio.stream.write_tum(a.out, traj)
    gt = drv.ground_truth()
    s = vio.stream.ape_stats(traj, gt, align=True)
    print("%d windows in %.1f ms (%.2f ms per window, host driver included); %s written" % (len(traj), dt * 1e3, dt * 1e3 / max(1, len(traj)), a.out))
    print("APE (-va): rmse %.4f m, mean %.4f, median %.4f, max %.4f over %d poses; unaligned rmse %.4f m"
          % (s["rmse"], s["mean"], s["median"], s["max"], s["pairs"], vio.stream.ate_rmse(traj, gt)))


if __name__ == "__main__":
    main()
