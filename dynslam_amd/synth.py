"""Deterministic KITTI-like synthetic input ("street canyon"), SURVEY.md 8(d).

No KITTI data exists in the container, so tests and bench feed the hot path with
an analytic scene rendered through the KITTI-odometry camera
(itm-sample-calib-from-kitti-odometry-sequence-06.txt:1-3) and quantised the way
the reference's depth providers do:

  depth -> disparity = f*b/z, b = 0.537150654273 (DynSLAMGUI.cpp:1185)
        -> + N(0, noise_px) -> rounded to 1/16 px (ELAS-like)
        -> depth int16 mm, kept only in [0.5 m, 20 m] else 0
           (DepthProvider.h:107-131, Input.h:71-72)

Camera frame: x right, y down, z forward.  Scene: ground plane y = +1.65 m,
facades at x = +-6 m with 0.3 m box relief every 4 m, car-sized boxes parked on
both sides.  Trajectory: 0.8 m/frame forward (10 Hz at 29 km/h) plus a small
yaw oscillation.  Optional moving boxes ("instances") come with their masks and
object poses for the per-instance volumes.
"""
import numpy as np

KITTI_FX = 707.0912
KITTI_FY = 707.0912
KITTI_CX = 601.8873
KITTI_CY = 183.1104
KITTI_BASELINE_M = 0.537150654273
MIN_DEPTH_M = 0.5
MAX_DEPTH_M = 20.0


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _ray_box(o, d, lo, hi):
    """Slab test; o (3,), d (...,3); returns entry t (inf where missed) and the hit axis."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (lo - o) * inv
        t1 = (hi - o) * inv
    tmin = np.minimum(t0, t1)
    tmax = np.maximum(t0, t1)
    tn = tmin.max(axis=-1)
    tf = tmax.min(axis=-1)
    hit = (tf >= tn) & (tn > 1e-6)
    axis = tmin.argmax(axis=-1)
    return np.where(hit, tn, np.inf), axis


class StreetScene:
    def __init__(self, width=1242, height=375, fx=None, fy=None, cx=None, cy=None, seed=1234,
                 noise_px=0.25, step_m=0.8, n_instances=0):
        sx = width / 1242.0
        sy = height / 375.0
        self.width, self.height = int(width), int(height)
        self.fx = float(fx if fx is not None else KITTI_FX * sx)
        self.fy = float(fy if fy is not None else KITTI_FY * sy)
        self.cx = float(cx if cx is not None else KITTI_CX * sx)
        self.cy = float(cy if cy is not None else KITTI_CY * sy)
        self.seed = int(seed)
        self.noise_px = float(noise_px)
        self.step_m = float(step_m)
        self.n_instances = int(n_instances)
        u = np.arange(self.width, dtype=np.float64)
        v = np.arange(self.height, dtype=np.float64)
        uu, vv = np.meshgrid(u, v)
        self._dcam = np.stack([(uu - self.cx) / self.fx, (vv - self.cy) / self.fy, np.ones_like(uu)], axis=-1)

    # -- trajectory ------------------------------------------------------------
    def pose(self, i):
        """camera->world 4x4 (the `invM` the host hands to SetPose)."""
        yaw = np.deg2rad(2.0) * np.sin(0.02 * i * 10.0)
        T = np.eye(4)
        T[:3, :3] = _rot_y(yaw)
        T[:3, 3] = [0.3 * np.sin(0.05 * i), 0.0, self.step_m * i]
        return T.astype(np.float32)

    def instance_pose(self, k, i):
        """object->world 4x4 of moving box k at frame i (constant velocity)."""
        T = np.eye(4)
        lane = -2.0 if (k % 2 == 0) else 2.0
        z0 = 8.0 + 3.5 * k
        # slower than the camera: gets overtaken; boxes 4.. (configs[3]: 7 instances) start beyond the 20 m
        # depth clip and are slow enough to come into range within ~15 frames
        speed = ((0.5 + 0.1 * k) if k < 4 else 0.3) * self.step_m
        T[:3, 3] = [lane, 1.65 - 0.75, z0 + speed * i]
        return T.astype(np.float32)

    # -- geometry --------------------------------------------------------------
    def _static_boxes(self, zc):
        boxes = []  # (lo, hi, id)
        boxes.append((np.array([-8.0, -8.0, -50.0]), np.array([-6.0, 1.65, 1e5]), 1))
        boxes.append((np.array([6.0, -8.0, -50.0]), np.array([8.0, 1.65, 1e5]), 2))
        k0 = int(np.floor((zc - 6.0) / 4.0))
        for k in range(k0, k0 + 9):
            z = 4.0 * k
            boxes.append((np.array([-6.0, -3.0, z]), np.array([-5.7, 1.65, z + 2.0]), 3))
            boxes.append((np.array([5.7, -3.0, z + 1.0]), np.array([6.0, 1.65, z + 3.0]), 4))
        c0 = int(np.floor((zc - 10.0) / 10.0))
        for c in range(c0, c0 + 5):  # parked cars, 4 x 1.6 x 1.5 m (l x w x h)
            z = 10.0 * c + 3.0
            side = -1.0 if (c % 2 == 0) else 1.0
            x0 = side * 4.4
            boxes.append((np.array([x0 - 0.8, 1.65 - 1.5, z]), np.array([x0 + 0.8, 1.65, z + 4.0]), 5 + (c % 4)))
        return boxes

    @staticmethod
    def _shade(pid, p, axis):
        """Procedural checker/stripe texture, uint8 RGB."""
        base = np.array([
            [150, 200, 255],  # sky
            [180, 120, 90], [170, 140, 110], [200, 200, 190], [190, 180, 160],
            [200, 40, 40], [40, 160, 60], [50, 80, 200], [220, 200, 40],
            [110, 110, 115],  # ground (id 9)
            [230, 120, 20], [20, 200, 200], [200, 60, 200], [240, 240, 240],
            [90, 60, 30], [60, 90, 30], [30, 60, 90],
        ], dtype=np.float64)
        col = base[np.clip(pid, 0, len(base) - 1)]
        chk = (np.floor(p[..., 0] * 2.0) + np.floor(p[..., 1] * 2.0) + np.floor(p[..., 2] * 2.0)).astype(np.int64) & 1
        stripe = (np.floor(p[..., 2] * 0.5).astype(np.int64) & 1)
        f = 0.75 + 0.25 * chk - 0.1 * stripe + 0.05 * axis
        return np.clip(col * f[..., None], 0, 255)

    def render(self, i, with_instances=True):
        """Exact depth (camera z, metres; inf = sky), RGB float image, primitive ids."""
        T = self.pose(i).astype(np.float64)
        R, o = T[:3, :3], T[:3, 3]
        d = self._dcam @ R.T
        best_t = np.full((self.height, self.width), np.inf)
        best_id = np.zeros((self.height, self.width), dtype=np.int64)
        best_axis = np.zeros((self.height, self.width), dtype=np.int64)
        # ground plane y = 1.65
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (1.65 - o[1]) / d[..., 1]
        tg = np.where((d[..., 1] > 1e-9) & (tg > 1e-6), tg, np.inf)
        best_t, best_id = tg, np.where(np.isfinite(tg), 9, 0)
        for lo, hi, pid in self._static_boxes(o[2]):
            t, ax = _ray_box(o, d, lo, hi)
            m = t < best_t
            best_t = np.where(m, t, best_t)
            best_id = np.where(m, pid, best_id)
            best_axis = np.where(m, ax, best_axis)
        inst_id = np.full((self.height, self.width), -1, dtype=np.int64)
        if with_instances:
            for k in range(self.n_instances):
                To = self.instance_pose(k, i).astype(np.float64)
                # boxes are axis aligned in the object frame; object frames are pure translations
                lo = To[:3, 3] + np.array([-0.8, -0.75, -2.0])
                hi = To[:3, 3] + np.array([0.8, 0.75, 2.0])
                t, ax = _ray_box(o, d, lo, hi)
                m = t < best_t
                best_t = np.where(m, t, best_t)
                best_id = np.where(m, 10 + (k % 7), best_id)
                best_axis = np.where(m, ax, best_axis)
                inst_id = np.where(m, k, inst_id)
        p = o + d * np.where(np.isfinite(best_t), best_t, 0.0)[..., None]
        rgb = self._shade(best_id, p, best_axis)
        return best_t, rgb, best_id, inst_id

    def frame(self, i, with_instances=True):
        """-> rgba uint8 [H,W,4], depth int16 mm [H,W], inv_m float32 4x4, inst_id int64 [H,W]."""
        z, rgb, _, inst_id = self.render(i, with_instances)
        rgba, depth_mm = self.quantise(i, z, rgb)
        return rgba, depth_mm, self.pose(i), inst_id

    def quantise(self, i, z, rgb):
        """Exact depth + float colour of frame i -> what a stereo matcher hands over: depth through a noisy
        disparity on a 1/16 px grid, clipped to [0.5, 20] m, int16 mm (0 = invalid); RGBA uint8."""
        rng = np.random.default_rng([self.seed, int(i)])
        fb = self.fx * KITTI_BASELINE_M
        with np.errstate(divide="ignore", invalid="ignore"):
            disp = np.where(np.isfinite(z), fb / z, 0.0)
        if self.noise_px > 0:
            disp = disp + rng.normal(0.0, self.noise_px, size=disp.shape)
        disp = np.round(disp * 16.0) / 16.0
        with np.errstate(divide="ignore", invalid="ignore"):
            zq = np.where(disp > 0, fb / disp, np.inf)
        ok = np.isfinite(z) & (zq >= MIN_DEPTH_M) & (zq <= MAX_DEPTH_M)
        depth_mm = np.where(ok, np.round(zq * 1000.0), 0.0).astype(np.int16)
        rgba = np.empty((self.height, self.width, 4), dtype=np.uint8)
        rgba[..., :3] = rgb.astype(np.uint8)
        rgba[..., 3] = 255
        return rgba, depth_mm

    def intrinsics(self):
        return self.fx, self.fy, self.cx, self.cy
