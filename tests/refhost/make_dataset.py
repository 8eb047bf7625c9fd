"""Writes a synthetic KITTI-odometry-style dataset in the on-disk layout the reference's input side reads
(Input.h:44-130 KittiOdometryDispnetConfig; PrecomputedSegmentationProvider.cpp:75-210), so that the
reference's whole per-frame pipeline can run on it (tests/refhost/ref_dynslam_host.cpp):

  image_2/%06d.ppm, image_3/%06d.ppm        left / right colour frames (binary PPM; cv::imread picks the decoder by signature)
  precomputed-depth-dispnet/%06d.pfm        disparity maps (float, bottom row first)
  seg_image_2/mnc/cls_%06d.png              segmentation preview (PPM content)
  seg_image_2/mnc/%06d.png.%04d.result.txt  "[x0 y0 x1 y1 0], probability, class" per detection
  seg_image_2/mnc/%06d.png.%04d.mask.txt    numpy text dump of the bbox-local 0/1 mask
  velodyne/%06d.bin                         "LIDAR" returns, float32 x y z reflectance in the camera frame (the host passes an
                                            identity velo-to-camera matrix): every 3rd pixel's EXACT surface point
  viso/%06d.bin                             what the scripted libviso2 stand-in "computes" for frame k >= 1: ego-motion,
                                            raw matches (object id in p_match::i1c), per-object motion vectors
  synthetic.txt                             W H fx fy cx cy baseline
  truth.npz                                 ground truth for the test's checks (not read by the host)

Scene: dynslam_amd.synth.StreetScene with three moving boxes whose speeds straddle the tracker's thresholds
(Track.h:96-98: < 0.03 m/frame static, > 0.55 dynamic, in between uncertain) and the parked cars of the street
as static detections.
"""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynslam_amd.synth import KITTI_BASELINE_M, StreetScene

CAR = 7  # Pascal VOC 2012 class id (SegmentationDataset.h:28-31)
OBJECT_SPEEDS = (0.30, 0.70, 1.00)  # m / frame along +z:  uncertain (cut away, never fused), dynamic, dynamic
PARKED_ID0 = 100


class PipelineScene(StreetScene):
    def __init__(self, width, height):
        super().__init__(width, height, n_instances=len(OBJECT_SPEEDS))

    def instance_pose(self, k, i):
        T = np.eye(4)
        lane = (-2.0, 2.0, 0.2)[k]
        z0 = (13.0, 11.0, 15.0)[k]
        T[:3, 3] = [lane, 1.65 - 0.75, z0 + OBJECT_SPEEDS[k] * i]
        return T.astype(np.float32)


def _rigid_to_vector(D):
    """4x4 rigid (rotation Rx*Ry*Rz, libviso2's convention) -> [rx, ry, rz, tx, ty, tz]."""
    R = D[:3, :3]
    ry = np.arcsin(np.clip(R[0, 2], -1.0, 1.0))
    rx = np.arctan2(-R[1, 2], R[2, 2])
    rz = np.arctan2(-R[0, 1], R[0, 0])
    return [float(rx), float(ry), float(rz), float(D[0, 3]), float(D[1, 3]), float(D[2, 3])]


def _write_ppm(path, rgb):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(np.ascontiguousarray(rgb, np.uint8).tobytes())


def _write_pfm(path, a):
    with open(path, "wb") as f:
        f.write(b"Pf\n%d %d\n-1.000000\n" % (a.shape[1], a.shape[0]))
        f.write(a[::-1].astype("<f4").tobytes())


def _interior(mask, margin):
    """mask eroded by `margin` pixels (4-neighbourhood, repeated)."""
    m = mask.copy()
    for _ in range(margin):
        e = m.copy()
        e[1:, :] &= m[:-1, :]
        e[:-1, :] &= m[1:, :]
        e[:, 1:] &= m[:, :-1]
        e[:, :-1] &= m[:, 1:]
        m = e
    return m


def write_dataset(root, n_frames, width=1242, height=375, min_area=45 * 45):
    sc = PipelineScene(width, height)
    fx, fy, cx, cy = sc.intrinsics()
    seg = os.path.join(root, "seg_image_2", "mnc")
    for d in ("image_2", "image_3", "precomputed-depth-dispnet", "viso", "csv", "velodyne"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    os.makedirs(seg, exist_ok=True)
    with open(os.path.join(root, "synthetic.txt"), "w") as f:
        f.write("%d %d %.9g %.9g %.9g %.9g %.12g\n" % (width, height, fx, fy, cx, cy, KITTI_BASELINE_M))
    truth = {"poses": [], "depth_mm": [], "detections": []}
    rng = np.random.default_rng(99)
    M_prev = None
    for i in range(n_frames):
        z, rgb, ids, inst_id = sc.render(i, with_instances=True)
        rgba, depth_mm = sc.quantise(i, z, rgb)
        T = sc.pose(i)
        M = np.linalg.inv(T.astype(np.float64))  # world -> camera
        _write_ppm(os.path.join(root, "image_2", "%06d.ppm" % i), rgba[..., :3])
        _write_ppm(os.path.join(root, "image_3", "%06d.ppm" % i), rgba[..., :3])
        with np.errstate(divide="ignore"):
            disp = np.where(depth_mm > 0, fx * KITTI_BASELINE_M / (depth_mm.astype(np.float64) / 1000.0), 0.0)
        _write_pfm(os.path.join(root, "precomputed-depth-dispnet", "%06d.pfm" % i), disp.astype(np.float32))
        _write_ppm(os.path.join(seg, "cls_%06d.png" % i), (rgba[..., :3] // 2))
        vv, uu = np.mgrid[1:height:3, 1:width:3]
        zz = z[vv, uu]
        keep = np.isfinite(zz) & (zz > 0.5) & (zz < 20.0)
        pts = np.stack([(uu[keep] - cx) / fx * zz[keep], (vv[keep] - cy) / fy * zz[keep], zz[keep], np.full(keep.sum(), 0.5)], axis=1)
        pts.astype("<f4").tofile(os.path.join(root, "velodyne", "%06d.bin" % i))

        # detections: the moving boxes, then the parked cars (prim ids 5..8 of the street)
        objects = [(k, inst_id == k, k) for k in range(sc.n_instances)]
        objects += [(PARKED_ID0 + pid, (ids == pid) & (inst_id < 0), None) for pid in (5, 6, 7, 8)]
        dets, motions, matches = [], [], []
        for oid, m, k in objects:
            ys, xs = np.nonzero(m)
            if len(ys) == 0:
                continue
            x0, x1, y0, y1 = int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())
            if (x1 - x0 + 1) * (y1 - y0 + 1) <= min_area or x0 <= 2 or x1 >= width - 3:
                continue  # too small for the provider (:103), or cut by the image border
            idx = len(dets)
            base = os.path.join(seg, "%06d.png.%04d" % (i, idx))
            with open(base + ".result.txt", "w") as f:
                f.write("[%d %d %d %d 0], %.3f, %d\n" % (x0, y0, x1, y1, 0.95, CAR))
            np.savetxt(base + ".mask.txt", m[y0:y1 + 1, x0:x1 + 1].astype(np.int32), fmt="%d")
            dets.append((oid, x0, y0, x1, y1))
            if M_prev is not None:
                if k is None:
                    D = M @ np.linalg.inv(M_prev)
                else:
                    O, Op = sc.instance_pose(k, i).astype(np.float64), sc.instance_pose(k, i - 1).astype(np.float64)
                    D = M @ O @ np.linalg.inv(Op) @ np.linalg.inv(M_prev)
                motions.append((oid, _rigid_to_vector(D)))
                iy, ix = np.nonzero(_interior(m, 4) & (depth_mm > 0))
                if len(iy) >= 18:
                    pick = rng.choice(len(iy), size=min(60, len(iy)), replace=False)
                    for y, x in zip(iy[pick], ix[pick]):
                        d = float(disp[y, x])
                        matches.append((float(x), float(y), 0, float(x) - d, float(y), 0, float(x), float(y), oid, float(x) - d, float(y), 0))
        if M_prev is not None:
            delta = M @ np.linalg.inv(M_prev)
            with open(os.path.join(root, "viso", "%06d.bin" % i), "wb") as f:
                f.write(np.ascontiguousarray(delta, np.float64).tobytes())
                f.write(struct.pack("<i", len(motions)))
                for oid, tr in motions:
                    f.write(struct.pack("<ii6d", oid, 1, *tr))
                f.write(struct.pack("<i", len(matches)))
                for mt in matches:
                    f.write(struct.pack("<ffiffiffiffi", *mt))
        M_prev = M
        truth["poses"].append(T)
        truth["depth_mm"].append(depth_mm)
        truth["detections"].append(dets)
    z_static = sc.render(n_frames - 1, with_instances=False)[0]  # what the last camera sees with the moving boxes taken away
    np.savez_compressed(os.path.join(root, "truth.npz"), poses=np.asarray(truth["poses"]), depth_mm=np.asarray(truth["depth_mm"]),
                        static_z_last=np.where(np.isfinite(z_static), z_static, 0.0).astype(np.float32), inst_last=inst_id.astype(np.int16))
    return truth


if __name__ == "__main__":
    write_dataset(sys.argv[1], int(sys.argv[2]))
