"""Seeded synthetic inputs for the registration path.

Scene   ground plane z=0 (+-100 m), four walls forming a 120 m x 80 m box,
        24 vertical cylinders (r=0.3 m, h=6 m), 12 yawed boxes 4x2x1.5 m.
Map     uniform-area samples of the scene, N(0, 1 cm) noise, float32.
Frame   HDL-64E model: 64 lasers, vertical angles linspace(-24.8, +2) deg in
        *sorted* order (raw laser ids follow HDLParser.cxx:179-181's beam LUT),
        1800 azimuth steps of 0.2 deg = 300 packets of 12 firing blocks
        (HDLParser.cxx:67-87 wire layout), ranges by ray casting from the moving
        sensor, N(0, 2 cm) range noise, 2 mm quantisation, >120 m dropped.
Motion  10 m/s along +x, yaw rate 5 deg/s, INS samples at 100 Hz.

This module only manufactures inputs (numpy); it computes nothing the product
is measured on.  tests/test_synth.py checks its packet decode against the
oracle's restatement of HDLParser.
"""
import math
import struct

import numpy as np

# HDLParser.cxx:179-181: sorted-by-vertical-angle index -> raw laser id
HDL64_BEAM_LUT = np.array(
    [38, 39, 42, 43, 32, 33, 36, 37, 40, 41, 46, 47, 50, 51, 54, 55, 44, 45, 48, 49, 52, 53, 58, 59,
     62, 63, 34, 35, 56, 57, 60, 61, 6, 7, 10, 11, 0, 1, 4, 5, 8, 9, 14, 15, 18, 19, 22, 23, 12, 13,
     16, 17, 20, 21, 26, 27, 30, 31, 2, 3, 24, 25, 28, 29], dtype=np.int32)

N_AZ = 1800
FIRINGS_PER_PKT = 12
AZ_PER_PKT = FIRINGS_PER_PKT // 2
PKTS_PER_FRAME = N_AZ // AZ_PER_PKT  # 300
FRAME_US = 100000
PKT_US = FRAME_US // PKTS_PER_FRAME  # 333
SENSOR_HEIGHT = 1.73
MAX_RANGE = 120.0


def hdl64_calibration(azimuth_correction=False):
    """(64, 9) float64 rows in HDLLaserCorrection order (HDLParser.cxx:89-100):
    azimuthCorrection, verticalCorrection, distanceCorrection,
    verticalOffsetCorrection, horizontalOffsetCorrection, sinVert, cosVert,
    sinVertOffset, cosVertOffset.  Units as after loadCorrectionsFile
    (:836-843: cm -> m)."""
    corr = np.zeros((64, 9))
    vert = np.linspace(-24.8, 2.0, 64)
    for i in range(64):
        raw = HDL64_BEAM_LUT[i]
        v = vert[i]
        corr[raw, 1] = v
        corr[raw, 2] = 0.0
        corr[raw, 3] = (0.20 if raw < 32 else 0.12)  # vertical offset, m
        corr[raw, 4] = 0.026 if (raw % 2 == 0) else -0.026
        if azimuth_correction:
            corr[raw, 0] = -4.5 + 9.0 * ((raw * 7) % 64) / 63.0
        rad = v * math.pi / 180.0
        corr[raw, 6] = math.cos(rad)
        corr[raw, 5] = math.sin(rad)
        corr[raw, 7] = corr[raw, 3] * corr[raw, 5]
        corr[raw, 8] = corr[raw, 3] * corr[raw, 6]
    return corr


_LUT = None


def rot_lut():
    """36001-entry cos/sin tables exactly as HDLParser.cxx:754-768 (libm, not numpy)."""
    global _LUT
    if _LUT is None:
        c = np.empty(36001)
        s = np.empty(36001)
        for i in range(36001):
            rad = (i / 100.0) * math.pi / 180.0
            c[i] = math.cos(rad)
            s[i] = math.sin(rad)
        _LUT = (c, s)
    return _LUT


def euler_matrix(roll_deg, pitch_deg, yaw_deg):
    """Ry(roll) Rx(pitch) Rz(yaw), the convention of PoseTransform::getMatrix
    (type_defs.h:134-146).  Generator-side only (float64 numpy)."""
    a, b, c = (math.radians(v) for v in (roll_deg, pitch_deg, yaw_deg))
    ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    rz = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
    return ry @ rx @ rz


class Scene:
    def __init__(self, seed=7):
        rng = np.random.default_rng(seed)
        self.ground_half = 100.0
        self.wall_x = 60.0
        self.wall_y = 40.0
        self.wall_h = 8.0
        # keep a corridor around the x axis free so the car can drive
        cyl = []
        while len(cyl) < 24:
            p = rng.uniform([-55, -36], [55, 36])
            if abs(p[1]) > 5.0:
                cyl.append(p)
        self.cyl = np.array(cyl)
        self.cyl_r = 0.3
        self.cyl_h = 6.0
        box = []
        while len(box) < 12:
            p = rng.uniform([-52, -33], [52, 33])
            if abs(p[1]) > 6.0:
                box.append([p[0], p[1], rng.uniform(0, 180)])
        self.box = np.array(box)
        self.box_half = np.array([2.0, 1.0, 0.75])

    # ------------------------------------------------------------ map sampling
    def sample_map(self, n, seed=1234, noise=0.01, dtype=np.float32):
        rng = np.random.default_rng(seed)
        g = self.ground_half
        areas = [4 * g * g,
                 2 * (2 * self.wall_x) * self.wall_h, 2 * (2 * self.wall_y) * self.wall_h,
                 24 * (2 * math.pi * self.cyl_r * self.cyl_h + math.pi * self.cyl_r ** 2),
                 12 * (2 * (4 * 1.5 + 2 * 1.5) + 4 * 2)]
        w = np.array(areas) / sum(areas)
        counts = np.floor(w * n).astype(np.int64)
        counts[0] += n - counts.sum()
        out = np.empty((n, 3), dtype=np.float64)
        nrm = np.empty((n, 3), dtype=np.float64)
        o = 0
        # ground
        k = counts[0]
        out[o:o + k, 0] = rng.uniform(-g, g, k)
        out[o:o + k, 1] = rng.uniform(-g, g, k)
        out[o:o + k, 2] = 0
        nrm[o:o + k] = [0, 0, 1]
        o += k
        # walls at x = +-wall_x (span y), then y = +-wall_y (span x)
        k = counts[1]
        sgn = rng.choice([-1.0, 1.0], k)
        out[o:o + k, 0] = sgn * self.wall_x
        out[o:o + k, 1] = rng.uniform(-self.wall_y, self.wall_y, k)
        out[o:o + k, 2] = rng.uniform(0, self.wall_h, k)
        nrm[o:o + k] = [1, 0, 0]
        o += k
        k = counts[2]
        sgn = rng.choice([-1.0, 1.0], k)
        out[o:o + k, 0] = rng.uniform(-self.wall_x, self.wall_x, k)
        out[o:o + k, 1] = sgn * self.wall_y
        out[o:o + k, 2] = rng.uniform(0, self.wall_h, k)
        nrm[o:o + k] = [0, 1, 0]
        o += k
        # cylinders: side + top cap
        k = counts[3]
        which = rng.integers(0, 24, k)
        side_a = 2 * math.pi * self.cyl_r * self.cyl_h
        cap_a = math.pi * self.cyl_r ** 2
        on_cap = rng.uniform(0, side_a + cap_a, k) > side_a
        ang = rng.uniform(0, 2 * math.pi, k)
        rad = np.where(on_cap, self.cyl_r * np.sqrt(rng.uniform(0, 1, k)), self.cyl_r)
        out[o:o + k, 0] = self.cyl[which, 0] + rad * np.cos(ang)
        out[o:o + k, 1] = self.cyl[which, 1] + rad * np.sin(ang)
        out[o:o + k, 2] = np.where(on_cap, self.cyl_h, rng.uniform(0, self.cyl_h, k))
        nrm[o:o + k, 0] = np.where(on_cap, 0, np.cos(ang))
        nrm[o:o + k, 1] = np.where(on_cap, 0, np.sin(ang))
        nrm[o:o + k, 2] = np.where(on_cap, 1, 0)
        o += k
        # boxes: 4 sides + top
        k = counts[4]
        which = rng.integers(0, 12, k)
        hx, hy, hz = self.box_half
        fa = np.array([2 * hy * 2 * hz, 2 * hy * 2 * hz, 2 * hx * 2 * hz, 2 * hx * 2 * hz,
                       2 * hx * 2 * hy])
        face = rng.choice(5, k, p=fa / fa.sum())
        u = rng.uniform(-1, 1, k)
        v = rng.uniform(-1, 1, k)
        lx = np.select([face == 0, face == 1], [hx, -hx], u * hx)
        ly = np.select([face == 2, face == 3], [hy, -hy], np.where(face < 2, u * hy, v * hy))
        lz = np.where(face == 4, 2 * hz, (v + 1) * hz)
        ln = np.zeros((k, 3))
        ln[face == 0] = [1, 0, 0]
        ln[face == 1] = [-1, 0, 0]
        ln[face == 2] = [0, 1, 0]
        ln[face == 3] = [0, -1, 0]
        ln[face == 4] = [0, 0, 1]
        yaw = np.radians(self.box[which, 2])
        cy, sy = np.cos(yaw), np.sin(yaw)
        out[o:o + k, 0] = self.box[which, 0] + cy * lx - sy * ly
        out[o:o + k, 1] = self.box[which, 1] + sy * lx + cy * ly
        out[o:o + k, 2] = lz
        nrm[o:o + k, 0] = cy * ln[:, 0] - sy * ln[:, 1]
        nrm[o:o + k, 1] = sy * ln[:, 0] + cy * ln[:, 1]
        nrm[o:o + k, 2] = ln[:, 2]
        o += k
        assert o == n
        out += nrm * rng.normal(0, noise, n)[:, None]
        perm = rng.permutation(n)
        out = out[perm]
        return (np.ascontiguousarray(out[:, 0], dtype=dtype),
                np.ascontiguousarray(out[:, 1], dtype=dtype),
                np.ascontiguousarray(out[:, 2], dtype=dtype))

    def sample_map_device(self, n, device, seed=1234, noise=0.01):
        """The same scene sampled with torch's generator on `device` (large maps: 10 M points
        take seconds in numpy, milliseconds on the GPU).  Same surfaces, same area weights, same
        noise model as sample_map, a different random stream -> a different but statistically
        identical point set.  Returns three float32 tensors (x, y, z) on `device`."""
        import torch
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed))
        f64 = torch.float64

        def uni(lo, hi, k):
            return lo + (hi - lo) * torch.rand(k, generator=gen, device=device, dtype=f64)

        def sign(k):
            return torch.randint(0, 2, (k,), generator=gen, device=device).to(f64) * 2.0 - 1.0

        g = self.ground_half
        areas = [4 * g * g,
                 2 * (2 * self.wall_x) * self.wall_h, 2 * (2 * self.wall_y) * self.wall_h,
                 24 * (2 * math.pi * self.cyl_r * self.cyl_h + math.pi * self.cyl_r ** 2),
                 12 * (2 * (4 * 1.5 + 2 * 1.5) + 4 * 2)]
        w = np.array(areas) / sum(areas)
        counts = np.floor(w * n).astype(np.int64)
        counts[0] += n - counts.sum()
        P, N = [], []
        k = int(counts[0])
        P.append(torch.stack([uni(-g, g, k), uni(-g, g, k), torch.zeros(k, device=device, dtype=f64)], 1))
        N.append(torch.tensor([0.0, 0.0, 1.0], device=device, dtype=f64).expand(k, 3))
        k = int(counts[1])
        P.append(torch.stack([sign(k) * self.wall_x, uni(-self.wall_y, self.wall_y, k), uni(0, self.wall_h, k)], 1))
        N.append(torch.tensor([1.0, 0.0, 0.0], device=device, dtype=f64).expand(k, 3))
        k = int(counts[2])
        P.append(torch.stack([uni(-self.wall_x, self.wall_x, k), sign(k) * self.wall_y, uni(0, self.wall_h, k)], 1))
        N.append(torch.tensor([0.0, 1.0, 0.0], device=device, dtype=f64).expand(k, 3))
        k = int(counts[3])
        cyl = torch.as_tensor(self.cyl, device=device, dtype=f64)
        which = torch.randint(0, 24, (k,), generator=gen, device=device)
        side_a = 2 * math.pi * self.cyl_r * self.cyl_h
        cap_a = math.pi * self.cyl_r ** 2
        on_cap = uni(0, side_a + cap_a, k) > side_a
        ang = uni(0, 2 * math.pi, k)
        rad = torch.where(on_cap, self.cyl_r * torch.sqrt(uni(0, 1, k)), torch.full_like(ang, self.cyl_r))
        zc = torch.where(on_cap, torch.full_like(ang, self.cyl_h), uni(0, self.cyl_h, k))
        P.append(torch.stack([cyl[which, 0] + rad * torch.cos(ang), cyl[which, 1] + rad * torch.sin(ang), zc], 1))
        zero = torch.zeros_like(ang)
        N.append(torch.stack([torch.where(on_cap, zero, torch.cos(ang)), torch.where(on_cap, zero, torch.sin(ang)),
                              on_cap.to(f64)], 1))
        k = int(counts[4])
        box = torch.as_tensor(self.box, device=device, dtype=f64)
        which = torch.randint(0, 12, (k,), generator=gen, device=device)
        hx, hy, hz = (float(v) for v in self.box_half)
        fa = torch.tensor([2 * hy * 2 * hz, 2 * hy * 2 * hz, 2 * hx * 2 * hz, 2 * hx * 2 * hz, 2 * hx * 2 * hy],
                          device=device, dtype=f64)
        face = torch.multinomial(fa / fa.sum(), k, replacement=True, generator=gen)
        u, v = uni(-1, 1, k), uni(-1, 1, k)
        lx = torch.where(face == 0, torch.full_like(u, hx), torch.where(face == 1, torch.full_like(u, -hx), u * hx))
        ly = torch.where(face == 2, torch.full_like(u, hy), torch.where(face == 3, torch.full_like(u, -hy),
                         torch.where(face < 2, u * hy, v * hy)))
        lz = torch.where(face == 4, torch.full_like(u, 2 * hz), (v + 1) * hz)
        lnx = (face == 0).to(f64) - (face == 1).to(f64)
        lny = (face == 2).to(f64) - (face == 3).to(f64)
        lnz = (face == 4).to(f64)
        yaw = torch.deg2rad(box[which, 2])
        cy, sy = torch.cos(yaw), torch.sin(yaw)
        P.append(torch.stack([box[which, 0] + cy * lx - sy * ly, box[which, 1] + sy * lx + cy * ly, lz], 1))
        N.append(torch.stack([cy * lnx - sy * lny, sy * lnx + cy * lny, lnz], 1))
        P, N = torch.cat(P), torch.cat(N)
        P = P + N * (noise * torch.randn(n, generator=gen, device=device, dtype=f64))[:, None]
        P = P[torch.randperm(n, generator=gen, device=device)].to(torch.float32)
        return P[:, 0].contiguous(), P[:, 1].contiguous(), P[:, 2].contiguous()

    # --------------------------------------------------------------- ray casting
    def raycast(self, o, d):
        """o, d: (n,3) float64 world-frame origins / unit directions -> hit distance (inf=miss)."""
        n = o.shape[0]
        best = np.full(n, np.inf)
        eps = 1e-9
        with np.errstate(divide="ignore", invalid="ignore"):
            # ground
            t = -o[:, 2] / d[:, 2]
            hx = o[:, 0] + t * d[:, 0]
            hy = o[:, 1] + t * d[:, 1]
            ok = (t > eps) & (np.abs(hx) <= self.ground_half) & (np.abs(hy) <= self.ground_half)
            best = np.where(ok & (t < best), t, best)
            # walls
            for sgn in (-1.0, 1.0):
                t = (sgn * self.wall_x - o[:, 0]) / d[:, 0]
                hy = o[:, 1] + t * d[:, 1]
                hz = o[:, 2] + t * d[:, 2]
                ok = (t > eps) & (np.abs(hy) <= self.wall_y) & (hz >= 0) & (hz <= self.wall_h)
                best = np.where(ok & (t < best), t, best)
                t = (sgn * self.wall_y - o[:, 1]) / d[:, 1]
                hx = o[:, 0] + t * d[:, 0]
                hz = o[:, 2] + t * d[:, 2]
                ok = (t > eps) & (np.abs(hx) <= self.wall_x) & (hz >= 0) & (hz <= self.wall_h)
                best = np.where(ok & (t < best), t, best)
            # cylinders (side only; caps are above the sensor's view mostly)
            a = d[:, 0] ** 2 + d[:, 1] ** 2
            for c in self.cyl:
                ox = o[:, 0] - c[0]
                oy = o[:, 1] - c[1]
                b = ox * d[:, 0] + oy * d[:, 1]
                cc = ox * ox + oy * oy - self.cyl_r ** 2
                disc = b * b - a * cc
                t = (-b - np.sqrt(np.where(disc >= 0, disc, np.nan))) / a
                hz = o[:, 2] + t * d[:, 2]
                ok = (disc >= 0) & (t > eps) & (hz >= 0) & (hz <= self.cyl_h)
                best = np.where(ok & (t < best), t, best)
            # yawed boxes via the slab test in the box frame
            for bx in self.box:
                yaw = math.radians(bx[2])
                cy, sy = math.cos(yaw), math.sin(yaw)
                rx = o[:, 0] - bx[0]
                ry = o[:, 1] - bx[1]
                lo = np.stack([cy * rx + sy * ry, -sy * rx + cy * ry, o[:, 2] - self.box_half[2]], 1)
                ld = np.stack([cy * d[:, 0] + sy * d[:, 1], -sy * d[:, 0] + cy * d[:, 1], d[:, 2]], 1)
                t1 = (-self.box_half - lo) / ld
                t2 = (self.box_half - lo) / ld
                tn = np.nanmax(np.minimum(t1, t2), axis=1)
                tf = np.nanmin(np.maximum(t1, t2), axis=1)
                ok = (tn <= tf) & (tn > eps)
                best = np.where(ok & (tn < best), tn, best)
        return best


class Motion:
    """Constant-speed, constant-yaw-rate drive."""

    def __init__(self, p0=(-30.0, 0.0, SENSOR_HEIGHT), speed=10.0, yaw0=0.0, yaw_rate=5.0,
                 t0_us=1_467_590_400_000_000):
        self.p0 = np.array(p0, dtype=np.float64)
        self.speed = speed
        self.yaw0 = yaw0
        self.yaw_rate = yaw_rate
        self.t0_us = t0_us

    def pose(self, t_us):
        """-> (T[3], Rdeg[3], V[3]) at absolute time t_us."""
        s = (t_us - self.t0_us) * 1e-6
        T = self.p0 + np.array([self.speed * s, 0.0, 0.0])
        R = np.array([0.0, 0.0, self.yaw0 + self.yaw_rate * s])
        V = np.array([self.speed, 0.0, 0.0])
        return T, R, V

    def ins_track(self, t_start_us, t_end_us, period_us=10000):
        """100 Hz INS samples covering [t_start, t_end] with one sample of margin."""
        k0 = (t_start_us - self.t0_us) // period_us - 1
        k1 = (t_end_us - self.t0_us) // period_us + 2
        out = []
        for k in range(k0, k1 + 1):
            t = self.t0_us + k * period_us
            T, R, V = self.pose(t)
            out.append((T, R, V, t))
        return out


def make_frame_packets(scene, motion, frame_idx, calib, seed=42, range_noise=0.02,
                       az_start=0):
    """One revolution as 300 raw 1206-byte packets.  Returns (packets, times_us, truth) where
    truth holds the per-packet true poses."""
    rng = np.random.default_rng(seed + frame_idx)
    cos_lut, sin_lut = rot_lut()
    t_frame = motion.t0_us + frame_idx * FRAME_US
    packets, times = [], []
    # ---- all rays of the frame at once: [pkt, firing(6), laser(64)]
    az_idx = (az_start + 20 * np.arange(N_AZ)) % 36000  # 0.2 deg steps in 0.01 deg units
    pkt_of_az = np.arange(N_AZ) // AZ_PER_PKT
    pkt_t = t_frame + pkt_of_az * PKT_US
    poses = [motion.pose(int(t_frame + p * PKT_US)) for p in range(PKTS_PER_FRAME)]
    Rw = np.stack([euler_matrix(*pz[1]) for pz in poses])  # (300,3,3)
    Tw = np.stack([pz[0] for pz in poses])  # (300,3)
    az_c = np.empty((N_AZ, 64))
    az_s = np.empty((N_AZ, 64))
    for l in range(64):
        if calib[l, 0] == 0:
            az_c[:, l] = cos_lut[az_idx]
            az_s[:, l] = sin_lut[az_idx]
        else:
            rad = ((az_idx / 100.0) - calib[l, 0]) * math.pi / 180.0
            az_c[:, l] = np.cos(rad)
            az_s[:, l] = np.sin(rad)
    cv, sv = calib[:, 6][None, :], calib[:, 5][None, :]
    hoff, voff = calib[:, 4][None, :], calib[:, 3][None, :]
    # sensor-frame ray: origin + d * dir  (HDLParser.cxx:614-623)
    dir_s = np.stack([cv * az_s, cv * az_c, np.broadcast_to(sv, az_s.shape)], -1)  # (1800,64,3)
    org_s = np.stack([-hoff * az_c, hoff * az_s, np.broadcast_to(voff, az_s.shape)], -1)
    R_az = Rw[pkt_of_az]  # (1800,3,3)
    T_az = Tw[pkt_of_az]
    dir_w = np.einsum("aij,alj->ali", R_az, dir_s).reshape(-1, 3)
    org_w = (np.einsum("aij,alj->ali", R_az, org_s) + T_az[:, None, :]).reshape(-1, 3)
    dist = scene.raycast(org_w, dir_w).reshape(N_AZ, 64)
    dist = dist + rng.normal(0, range_noise, dist.shape)
    dist = dist - calib[:, 2][None, :]
    raw = np.where(np.isfinite(dist) & (dist < MAX_RANGE) & (dist > 0.9),
                   np.rint(dist / 0.002), 0).astype(np.int64)
    raw = np.clip(raw, 0, 65535).astype(np.uint16)
    inten = rng.integers(1, 255, raw.shape, dtype=np.uint8)
    for p in range(PKTS_PER_FRAME):
        buf = bytearray(1206)
        for f in range(FIRINGS_PER_PKT):
            a = p * AZ_PER_PKT + f // 2
            upper = (f % 2 == 0)
            struct.pack_into("<HH", buf, 100 * f, 0xEEFF if upper else 0xDDFF, int(az_idx[a]))
            l0 = 0 if upper else 32
            blk = np.empty(32, dtype=[("d", "<u2"), ("i", "u1")])
            blk["d"] = raw[a, l0:l0 + 32]
            blk["i"] = inten[a, l0:l0 + 32]
            buf[100 * f + 4:100 * f + 100] = blk.tobytes()
        t_pkt = int(t_frame + p * PKT_US)
        struct.pack_into("<I", buf, 1200, (t_pkt % 3_600_000_000) & 0xFFFFFFFF)
        packets.append(bytes(buf))
        times.append(t_pkt)
    truth = dict(R=Rw, T=Tw, t_frame=t_frame)
    return packets, times, truth


def decode_sensor_frame(packets, calib):
    """Numpy decode of one frame's packets WITHOUT a geo-transform: the 'decoded HDLFrame'
    that is K1's input.  Returns beam-major (HDLFrame::getPointsAsOneCloud order,
    HDLFrame.cxx:127-144, after the beam-LUT permutation of HDLParser.cxx:880-893)
    float32 x,y,z,intensity, uint16 packet index, int32 beam_start[65].
    Assumes the packets hold exactly one revolution (no split inside)."""
    cos_lut, sin_lut = rot_lut()
    npk = len(packets)
    arr = np.frombuffer(b"".join(packets), dtype=np.uint8).reshape(npk, 1206)
    fire = arr[:, :1200].reshape(npk, 12, 100)
    ident = fire[:, :, 0].astype(np.uint16) | (fire[:, :, 1].astype(np.uint16) << 8)
    rot = (fire[:, :, 2].astype(np.uint16) | (fire[:, :, 3].astype(np.uint16) << 8)) % 36000
    ret = fire[:, :, 4:].reshape(npk, 12, 32, 3)
    dist_raw = ret[..., 0].astype(np.uint16) | (ret[..., 1].astype(np.uint16) << 8)
    inten = ret[..., 2]
    offset = np.where(ident == 0xEEFF, 0, 32)  # (npk,12)
    laser = offset[:, :, None] + np.arange(32)[None, None, :]  # raw laser id
    az = np.broadcast_to(rot[:, :, None], laser.shape)
    c = calib[laser]  # (npk,12,32,9)
    use_lut = c[..., 0] == 0
    rad = ((az.astype(np.float64) / 100.0) - c[..., 0]) * math.pi / 180.0
    cos_az = np.where(use_lut, cos_lut[az], np.cos(rad))
    sin_az = np.where(use_lut, sin_lut[az], np.sin(rad))
    dm = dist_raw * 0.002 + c[..., 2]
    xy = dm * c[..., 6]
    px = xy * sin_az - c[..., 4] * cos_az
    py = xy * cos_az + c[..., 4] * sin_az
    pz = dm * c[..., 5] + c[..., 3]
    valid = dist_raw != 0
    pkt = np.broadcast_to(np.arange(npk)[:, None, None], laser.shape)
    inv_lut = np.empty(64, dtype=np.int32)
    inv_lut[HDL64_BEAM_LUT] = np.arange(64)  # raw laser id -> sorted beam index
    beam = inv_lut[laser]
    sel = valid.ravel()
    beam_f = beam.ravel()[sel]
    order = np.argsort(beam_f, kind="stable")  # beam-major, firing order preserved within
    def take(a, dt):
        return np.ascontiguousarray(a.ravel()[sel][order], dtype=dt)
    beam_start = np.zeros(65, dtype=np.int32)
    beam_start[1:] = np.cumsum(np.bincount(beam_f, minlength=64))
    return dict(x=take(px, np.float32), y=take(py, np.float32), z=take(pz, np.float32),
                intensity=take(inten, np.float32), pkt=take(pkt, np.uint16),
                azimuth=take(az, np.uint16), distance=take(dm, np.float32),
                beam_start=beam_start)


def perturbed_guess(T_true, dt=(0.30, -0.20, 0.05), drot_deg=(0.5, -0.3, 1.0)):
    """SURVEY 8d initial guess: truth perturbed by dt metres and roll/pitch/yaw degrees.
    T_true is a 3x4 row-major (12,) matrix; the rotation perturbation acts about the
    frame origin (left-multiplied on the rotation part only)."""
    M = np.array(T_true, dtype=np.float64).reshape(3, 4).copy()
    Rp = euler_matrix(*drot_deg)
    M[:, :3] = Rp @ M[:, :3]
    M[:, 3] += np.array(dt)
    return M.reshape(12)


# ------------------------------------------------------------------------------------------------
# A world long enough to DRIVE through (BASELINE configs[2] measured as mapping: the map starts from
# the first frame and grows only from accepted increments, so the car has to keep seeing new ground).
class LongScene:
    """A street: ground |y| <= 100 m for x in [-100, length + 100]; side walls y = +-40 m (8 m high) all
    along it; every 50 m a pair of fins (cross walls, 12 m into the street from either side wall: what
    constrains x); per 100 m block 24 cylinders and 12 yawed boxes, seeded per block, clear of the lane.
    Same surface kinds and sizes as Scene, a different arrangement; ray casting is written once for numpy
    and torch (`xp`), culled to what lies within the sensor's range of the ray origins."""

    def __init__(self, length=800.0, seed=7):
        self.length = float(length)
        self.x_lo, self.x_hi = -100.0, self.length + 100.0
        self.ground_half = 100.0
        self.wall_y, self.wall_h = 40.0, 8.0
        self.fin_depth, self.fin_every = 12.0, 50.0
        self.fin_x = np.arange(self.x_lo + 25.0, self.x_hi, self.fin_every)
        self.cyl_r, self.cyl_h = 0.3, 6.0
        self.box_half = np.array([2.0, 1.0, 0.75])
        cyl, box = [], []
        nb = int(math.ceil((self.x_hi - self.x_lo) / 100.0))
        for b in range(nb):
            rng = np.random.default_rng(seed * 1000 + b)
            x0 = self.x_lo + 100.0 * b
            k = 0
            while k < 24:
                p = rng.uniform([x0, -36], [x0 + 100.0, 36])
                if abs(p[1]) > 5.0:
                    cyl.append(p)
                    k += 1
            k = 0
            while k < 12:
                p = rng.uniform([x0 + 3, -33], [x0 + 97.0, 33])
                if abs(p[1]) > 6.0:
                    box.append([p[0], p[1], rng.uniform(0, 180)])
                    k += 1
        self.cyl = np.array(cyl)
        self.box = np.array(box)

    def raycast(self, o, d, xp=np):
        """o, d: (n, 3) float64 origins / unit directions (numpy arrays or torch tensors, `xp` the module) ->
        hit distance (inf = miss)."""
        is_np = xp is np
        ox, oy, oz = o[:, 0], o[:, 1], o[:, 2]
        dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
        inf = float("inf")
        best = xp.full_like(ox, inf)
        eps = 1e-9
        cx = float(ox.mean()) if is_np else float(ox.mean().item())   # (one revolution: the origins are ~1 m apart)
        reach = MAX_RANGE + 5.0

        def take(best, t, ok):
            return xp.where(ok & (t < best), t, best)

        ctxm = np.errstate(divide="ignore", invalid="ignore") if is_np else _NullCtx()
        with ctxm:
            t = -oz / dz
            hx, hy = ox + t * dx, oy + t * dy
            best = take(best, t, (t > eps) & (hx >= self.x_lo) & (hx <= self.x_hi) & (xp.abs(hy) <= self.ground_half))
            for sgn in (-1.0, 1.0):
                t = (sgn * self.wall_y - oy) / dy
                hx, hz = ox + t * dx, oz + t * dz
                best = take(best, t, (t > eps) & (hx >= self.x_lo) & (hx <= self.x_hi) & (hz >= 0) & (hz <= self.wall_h))
            for fx in self.fin_x[np.abs(self.fin_x - cx) <= reach]:
                t = (float(fx) - ox) / dx
                hy, hz = oy + t * dy, oz + t * dz
                ay = xp.abs(hy)
                best = take(best, t, (t > eps) & (ay <= self.wall_y) & (ay >= self.wall_y - self.fin_depth) &
                            (hz >= 0) & (hz <= self.wall_h))
            a = dx * dx + dy * dy
            for c in self.cyl[np.abs(self.cyl[:, 0] - cx) <= reach]:
                rx, ry = ox - float(c[0]), oy - float(c[1])
                b = rx * dx + ry * dy
                cc = rx * rx + ry * ry - self.cyl_r ** 2
                disc = b * b - a * cc
                t = (-b - xp.sqrt(xp.where(disc >= 0, disc, xp.full_like(disc, float("nan"))))) / a
                hz = oz + t * dz
                best = take(best, t, (disc >= 0) & (t > eps) & (hz >= 0) & (hz <= self.cyl_h))
            hxb, hyb, hzb = (float(v) for v in self.box_half)
            big = 1.0e300
            for bx in self.box[np.abs(self.box[:, 0] - cx) <= reach]:
                yaw = math.radians(bx[2])
                cy, sy = math.cos(yaw), math.sin(yaw)
                rx, ry = ox - float(bx[0]), oy - float(bx[1])
                lo3 = (cy * rx + sy * ry, -sy * rx + cy * ry, oz - hzb)
                ld3 = (cy * dx + sy * dy, -sy * dx + cy * dy, dz)
                tn = xp.full_like(ox, -big)
                tf = xp.full_like(ox, big)
                for lo_, ld_, h_ in zip(lo3, ld3, (hxb, hyb, hzb)):
                    t1, t2 = (-h_ - lo_) / ld_, (h_ - lo_) / ld_
                    # (a ray parallel to the slab: 0/0 -> nan compares false everywhere: the slab is ignored, as nanmax did)
                    lo_t, hi_t = xp.minimum(t1, t2), xp.maximum(t1, t2)
                    tn = xp.where(lo_t > tn, lo_t, tn)
                    tf = xp.where(hi_t < tf, hi_t, tf)
                best = take(best, tn, (tn <= tf) & (tn > eps))
        return best


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def make_frame_packets_device(scene, motion, frame_indices, calib, device, seed=42, range_noise=0.02):
    """make_frame_packets for many frames with the ray casting and the packet assembly in torch on `device`
    (600 revolutions take seconds instead of minutes).  Same sensor model, wire layout and noise model; another
    random stream.  -> (uint8 tensor [n_frames, 300, 1206] on `device`, int64 numpy times [n_frames, 300])."""
    import torch
    f64 = torch.float64
    cos_lut, sin_lut = rot_lut()
    az_idx = (20 * np.arange(N_AZ)) % 36000
    pkt_of_az = np.arange(N_AZ) // AZ_PER_PKT
    az_c = np.empty((N_AZ, 64))
    az_s = np.empty((N_AZ, 64))
    for l in range(64):
        if calib[l, 0] == 0:
            az_c[:, l] = cos_lut[az_idx]
            az_s[:, l] = sin_lut[az_idx]
        else:
            rad = ((az_idx / 100.0) - calib[l, 0]) * math.pi / 180.0
            az_c[:, l] = np.cos(rad)
            az_s[:, l] = np.sin(rad)
    cv, sv = calib[:, 6][None, :], calib[:, 5][None, :]
    hoff, voff = calib[:, 4][None, :], calib[:, 3][None, :]
    dir_s = torch.as_tensor(np.stack([cv * az_s, cv * az_c, np.broadcast_to(sv, az_s.shape)], -1), device=device, dtype=f64)
    org_s = torch.as_tensor(np.stack([-hoff * az_c, hoff * az_s, np.broadcast_to(voff, az_s.shape)], -1), device=device, dtype=f64)
    dcorr = torch.as_tensor(calib[:, 2][None, :].copy(), device=device, dtype=f64)
    az_t = torch.as_tensor(az_idx, device=device, dtype=torch.int64)
    gen = torch.Generator(device=device)
    out, times = [], []
    hdr = torch.empty((N_AZ, 2, 4), dtype=torch.uint8, device=device)
    hdr[:, :, 0] = 0xFF
    hdr[:, 0, 1] = 0xEE
    hdr[:, 1, 1] = 0xDD
    hdr[:, :, 2] = (az_t & 0xFF).to(torch.uint8)[:, None]
    hdr[:, :, 3] = (az_t >> 8).to(torch.uint8)[:, None]
    for fi in frame_indices:
        t_frame = motion.t0_us + int(fi) * FRAME_US
        poses = [motion.pose(int(t_frame + p * PKT_US)) for p in range(PKTS_PER_FRAME)]
        Rw = torch.as_tensor(np.stack([euler_matrix(*pz[1]) for pz in poses])[pkt_of_az], device=device, dtype=f64)
        Tw = torch.as_tensor(np.stack([pz[0] for pz in poses])[pkt_of_az], device=device, dtype=f64)
        dir_w = torch.einsum("aij,alj->ali", Rw, dir_s).reshape(-1, 3)
        org_w = (torch.einsum("aij,alj->ali", Rw, org_s) + Tw[:, None, :]).reshape(-1, 3)
        dist = scene.raycast(org_w, dir_w, xp=torch).reshape(N_AZ, 64)
        gen.manual_seed(int(seed) * 100003 + int(fi))
        dist = dist + range_noise * torch.randn(dist.shape, generator=gen, device=device, dtype=f64) - dcorr
        ok = torch.isfinite(dist) & (dist < MAX_RANGE) & (dist > 0.9)
        raw = torch.where(ok, torch.round(dist / 0.002), torch.zeros_like(dist)).clamp(0, 65535).to(torch.int64)
        inten = torch.randint(1, 255, raw.shape, generator=gen, device=device, dtype=torch.int64)
        body = torch.stack([(raw & 0xFF), (raw >> 8), inten], -1).to(torch.uint8).reshape(N_AZ, 2, 96)
        blocks = torch.cat([hdr, body], 2).reshape(PKTS_PER_FRAME, 1200)
        t_pkt = t_frame + np.arange(PKTS_PER_FRAME, dtype=np.int64) * PKT_US
        stamp = torch.as_tensor(((t_pkt % 3_600_000_000) & 0xFFFFFFFF).astype(np.int64), device=device)
        tail = torch.stack([(stamp >> s) & 0xFF for s in (0, 8, 16, 24)] + [torch.zeros_like(stamp)] * 2, 1).to(torch.uint8)
        out.append(torch.cat([blocks, tail], 1))
        times.append(t_pkt)
    return torch.stack(out), np.stack(times)
