// decode.hip -- SURVEY row f1: Velodyne packet decode + per-laser calibration + frame split
// + per-packet motion compensation, on the GPU, from the raw 1206-byte packets
// (HDLParser.cxx:67-87 wire layout).  Replaces the reference's single-threaded hot loop
// processHDLPacket -> processFiring -> pushFiringData (HDLParser.cxx:980-1055, 900-977,
// 587-752) and splitFrame's beam re-ordering (:867-897).  Semantics == oracle/decode.c,
// bit for bit (fp64 polar->Cartesian with the host-built sin/cos tables, the affine of
// type_defs.h:160-166 with separately rounded products, one rounding to f32).
//
// The sequential part of the parser (which firing block belongs to which frame, where a
// packet starts after a split, the car pose of a frame) is 12 integers per packet and is
// planned on the host (capi.cpp: plan_decode); the per-return arithmetic and the beam-major
// compaction run here:
//   k_decode_keys   one thread per return: validity + key = frame*64 + output beam
//   (stable radix sort of (key, return index): beam-major, firing order inside a beam)
//   k_decode_emit   one thread per surviving return: decode, calibrate, compensate, store
#include "device_math.hpp"

namespace velo {

struct Ret {
    bool ok;
    int frame, beam, laser;
    unsigned short azimuth, raw_dist;
    unsigned char intensity;
    int pkt;
};

__device__ __forceinline__ double hdl32_adjust(int block, int dsr) { return (block * 46.08) + (dsr * 1.152); }
__device__ __forceinline__ double vlp16_adjust(int block, int dsr, int within)
{
    return (block * 110.592) + (dsr * 2.304) + (within * 55.296);
}

// HDLParser.cxx:900-977 for one (packet, block, dsr)
__device__ __forceinline__ Ret classify(const DecodeView& v, size_t r)
{
    Ret o;
    o.ok = false;
    const int dsr = (int)(r & 31);
    const int blk = (int)((r >> 5) % 12);
    const int pkt = (int)(r / 384);
    o.pkt = pkt;
    const int fr = v.blk_frame[(size_t)pkt * 12 + blk];
    if (fr < 0) return o;
    const uint8_t* fd = v.pkts + (size_t)pkt * 1206 + 100 * blk;
    const unsigned id = fd[0] | (fd[1] << 8);
    const unsigned short rot = (unsigned short)(fd[2] | (fd[3] << 8));
    const int offset = (id == 0xeeff) ? 0 : 32;
    int laser = dsr + offset;
    int within = 0;
    if (v.n_lasers == 16 && laser >= 16) {
        laser -= 16;
        within = 1;
    }
    double ts_adj = 0.0, blk0 = 0.0, nblk0 = 1.0;
    if (v.n_lasers == 32) {
        ts_adj = hdl32_adjust(blk, dsr);
        nblk0 = hdl32_adjust(blk + 1, 0);
        blk0 = hdl32_adjust(blk, 0);
    } else if (v.n_lasers == 16) {
        ts_adj = vlp16_adjust(blk, laser, within);
        nblk0 = vlp16_adjust(blk + 1, 0, 0);
        blk0 = vlp16_adjust(blk, 0, 0);
    }
    const int az_adj = (int)round((double)v.az_diff[pkt] * ((ts_adj - blk0) / (nblk0 - blk0)));
    const uint8_t* lr = fd + 4 + 3 * dsr;
    o.raw_dist = (unsigned short)(lr[0] | (lr[1] << 8));
    o.intensity = lr[2];
    if (o.raw_dist == 0) return o;
    if (laser >= v.n_lasers && v.n_lasers < 64) return o;  // no such beam in the frame
    if (!((v.laser_mask >> laser) & 1ull)) return o;       // de-selected laser (:964)
    o.laser = laser;
    o.azimuth = (unsigned short)((unsigned short)(rot + az_adj) % 36000);
    o.frame = fr;
    o.beam = v.frame_perm[fr] ? v.inv_lut[laser] : laser;
    o.ok = true;
    return o;
}

// HDLParser.cxx:587-640: polar -> Cartesian with the laser's corrections
__device__ __forceinline__ void raw_position(const DecodeView& v, const Ret& r, int corr_idx,
                                             double pos[3], double& distance_m)
{
    const double* c = v.corr + 9 * (size_t)corr_idx;
    double cos_az, sin_az;
    if (c[0] == 0) {
        cos_az = v.lut_cos[r.azimuth];
        sin_az = v.lut_sin[r.azimuth];
    } else {
        cos_az = v.az_cos[(size_t)corr_idx * 36000 + r.azimuth];
        sin_az = v.az_sin[(size_t)corr_idx * 36000 + r.azimuth];
    }
    distance_m = r.raw_dist * 0.002 + c[2];
    const double xy = distance_m * c[6];
    pos[0] = xy * sin_az - c[4] * cos_az;
    pos[1] = xy * cos_az + c[4] * sin_az;
    pos[2] = distance_m * c[5] + c[3];
}

__device__ __forceinline__ bool cropped(const DecodeView& v, const double pos[3])
{
    if (!v.crop) return false;
    const bool in_box = pos[0] >= v.region[0] && pos[0] <= v.region[1] && pos[1] >= v.region[2] &&
                        pos[1] <= v.region[3] && pos[2] >= v.region[4] && pos[2] <= v.region[5];
    return (in_box && !v.crop_inside) || (!in_box && v.crop_inside);  // :629-639
}

__global__ __launch_bounds__(256) void k_decode_keys(DecodeView v, size_t n_ret,
                                                     uint32_t* __restrict__ keys,
                                                     uint32_t* __restrict__ idx)
{
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_ret;
         r += (size_t)gridDim.x * blockDim.x) {
        const Ret o = classify(v, r);
        uint32_t key = 0xffffffffu;
        if (o.ok) {
            bool keep = true;
            if (v.crop) {
                double pos[3], dm;
                const int dsr = (int)(r & 31);
                const uint8_t* fd = v.pkts + (size_t)o.pkt * 1206 + 100 * (int)((r >> 5) % 12);
                const int offset = ((fd[0] | (fd[1] << 8)) == 0xeeff) ? 0 : 32;
                raw_position(v, o, dsr + offset, pos, dm);
                keep = !cropped(v, pos);
            }
            if (keep) key = (uint32_t)o.frame * 64u + (uint32_t)o.beam;
        }
        keys[r] = key;
        idx[r] = (uint32_t)r;
    }
}

hipError_t launch_decode_keys(const DecodeView& v, size_t n_ret, uint32_t* keys, uint32_t* idx,
                              hipStream_t s)
{
    if (n_ret == 0) return hipSuccess;
    size_t g = (n_ret + 255) / 256;
    hipLaunchKernelGGL(k_decode_keys, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, s, v, n_ret,
                       keys, idx);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_decode_emit(DecodeView v, const uint32_t* __restrict__ order,
                                                     size_t n_valid, float* __restrict__ ox,
                                                     float* __restrict__ oy, float* __restrict__ oz,
                                                     float* __restrict__ oi,
                                                     uint16_t* __restrict__ oaz,
                                                     float* __restrict__ odist,
                                                     uint16_t* __restrict__ opkt)
{
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_valid;
         s += (size_t)gridDim.x * blockDim.x) {
        const size_t r = order[s];
        const Ret o = classify(v, r);
        const int dsr = (int)(r & 31);
        const uint8_t* fd = v.pkts + (size_t)o.pkt * 1206 + 100 * (int)((r >> 5) % 12);
        const int offset = ((fd[0] | (fd[1] << 8)) == 0xeeff) ? 0 : 32;
        double pos[3], dm;
        raw_position(v, o, dsr + offset, pos, dm);
        if (v.tvalid[o.pkt]) {  // type_defs.h:160-166
            const double* M = v.table + 12 * (size_t)o.pkt;
            const double x = pos[0], y = pos[1], z = pos[2];
            pos[0] = M[0] * x + M[1] * y + M[2] * z + M[3];
            pos[1] = M[4] * x + M[5] * y + M[6] * z + M[7];
            pos[2] = M[8] * x + M[9] * y + M[10] * z + M[11];
        }
        ox[s] = (float)pos[0];
        oy[s] = (float)pos[1];
        oz[s] = (float)pos[2];
        oi[s] = (float)(short)o.intensity;
        oaz[s] = o.azimuth;
        odist[s] = (float)dm;
        opkt[s] = (uint16_t)min(o.pkt, 65535);
    }
}

hipError_t launch_decode_emit(const DecodeView& v, const uint32_t* order, size_t n_valid, float* ox,
                              float* oy, float* oz, float* oi, uint16_t* oaz, float* odist,
                              uint16_t* opkt, hipStream_t s)
{
    if (n_valid == 0) return hipSuccess;
    size_t g = (n_valid + 255) / 256;
    hipLaunchKernelGGL(k_decode_emit, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, s, v, order,
                       n_valid, ox, oy, oz, oi, oaz, odist, opkt);
    return hipGetLastError();
}

// starts[k] = number of sorted keys < k, k in [0, n_keys]
__global__ __launch_bounds__(256) void k_key_starts(const uint32_t* __restrict__ keys, size_t n,
                                                    uint32_t n_keys, int32_t* __restrict__ starts)
{
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k <= n_keys;
         k += gridDim.x * blockDim.x) {
        size_t lo = 0, hi = n;
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if (keys[mid] < k)
                lo = mid + 1;
            else
                hi = mid;
        }
        starts[k] = (int32_t)lo;
    }
}

hipError_t launch_key_starts(const uint32_t* keys, size_t n, uint32_t n_keys, int32_t* starts,
                             hipStream_t s)
{
    const uint32_t g = (n_keys + 1 + 255) / 256;
    hipLaunchKernelGGL(k_key_starts, dim3(g > 1024 ? 1024 : g), dim3(256), 0, s, keys, n, n_keys,
                       starts);
    return hipGetLastError();
}

}  // namespace velo
