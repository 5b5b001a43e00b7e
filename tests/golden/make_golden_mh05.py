#!/usr/bin/env python3
"""A stretch of the only real sensor data the reference ships: EuRoC MH_05 IMU samples and camera stamps
(VM/config/MH_05_imu0.txt: stamp[ns] gyro[rad/s] acc[m/s^2] at 200 Hz; VM/config/MH_05_cam0.txt: stamp[ns] file, 20 Hz),
read the way VM/test/run_euroc.cpp:26-76 reads them (`>> double`, then / 1e9), with the sensor parameters of
VM/config/euroc_config.yaml.  DATA only (numbers the reference's own runner consumes), no source text.

    python tests/golden/make_golden_mh05.py      (needs /root/reference)  ->  tests/golden/mh05_imu_stretch.npz

Frames: every second camera stamp (the tracker publishes at `freq: 10` Hz, euroc_config.yaml), N_FRAMES of them from camera line
FIRST_CAM on (20 s into the sequence: the MAV is flying)."""
import os
import re

import numpy as np

VM = "/root/reference/workspace/assignments/17-vins-initialization/vins-mono"
HERE = os.path.dirname(os.path.abspath(__file__))
FIRST_CAM, N_FRAMES, STEP = 400, 36, 2

cam = [float(line.split()[0]) / 1e9 for line in open(os.path.join(VM, "config", "MH_05_cam0.txt")) if line.strip()]
cam_t = np.array(cam[FIRST_CAM:FIRST_CAM + STEP * N_FRAMES:STEP])
rows = []
for line in open(os.path.join(VM, "config", "MH_05_imu0.txt")):
    p = line.split()
    if len(p) == 7:
        rows.append([float(p[0]) / 1e9] + [float(v) for v in p[1:]])
imu = np.array(rows)
keep = (imu[:, 0] >= cam_t[0] - 0.011) & (imu[:, 0] <= cam_t[-1] + 0.011)
imu = imu[keep]

yaml = open(os.path.join(VM, "config", "euroc_config.yaml")).read()


def scalar(name):
    return float(re.search(r"^%s:\s*([-+0-9.eE]+)" % name, yaml, re.M).group(1))


def matrix(name, n):
    m = re.search(r"%s:.*?data:\s*\[(.*?)\]" % name, yaml, re.S)
    return np.array([float(v) for v in m.group(1).replace("\n", " ").split(",")]).reshape(n)


np.savez_compressed(os.path.join(HERE, "mh05_imu_stretch.npz"),
                    imu_t=imu[:, 0], imu_gyr=imu[:, 1:4], imu_acc=imu[:, 4:7], cam_t=cam_t,
                    acc_n=scalar("acc_n"), gyr_n=scalar("gyr_n"), acc_w=scalar("acc_w"), gyr_w=scalar("gyr_w"), g_norm=scalar("g_norm"),
                    ric=matrix("extrinsicRotation", (3, 3)), tic=matrix("extrinsicTranslation", (3,)))
print("frames %d (%.3f s), imu samples %d, median dt %.6f s" % (len(cam_t), cam_t[-1] - cam_t[0], len(imu), np.median(np.diff(imu[:, 0]))))
