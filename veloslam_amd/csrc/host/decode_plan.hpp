// decode_plan.hpp -- the HOST half of the packet decode: the sequential part of the
// reference's parser -- HDLParser::processHDLPacket (HDLParser.cxx:980-1055): the pose of every packet
// and its 3x4 table relative to the frame's first pose, the frame split on an azimuth wrap, which
// frame each of a packet's 12 firing blocks belongs to, the median azimuth step -- run over
// [packets still in flight] + [new packets], with everything the device half (kernels/decode.hip:
// the per-return arithmetic) needs laid out in ONE staging buffer.  Plain host code: no GPU type, no
// GPU call; the staging buffer comes from an allocator the caller supplies (pinned memory in the
// library, malloc in the CPU tests), so this half is built with g++ like the rest of host/ and runs
// under the sanitizers (tests/cpp/host_fuzz.cpp) and against the oracle parser without a GPU
// (tests/cpp/plan_dump.cpp).
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../../include/velo.h"
#include "../../../include/veloslam/PoseTransform.hpp"

namespace velo {

// parser state carried across velo_decode_stream calls (HDLParser is stateful across packets)
struct DecodeStream {
    int last_az = -1, firing_skip = 0;
    bool inited = false, is_hdl64 = false, open = false;
    veloslam::PoseTransform carpose;
    velo_pose carpose0{};      // header of the unfinished frame
    int64_t frame_t = VELO_TIME_INVALID;
    int32_t frame_packets = 0;
    // the packets that still hold firing blocks of the unfinished frame
    std::vector<uint8_t> bytes, tvalid;
    std::vector<int64_t> t;
    std::vector<double> table;
    std::vector<int32_t> azdiff;
    std::vector<int16_t> blk;  // 12 per packet: 0 = unfinished frame, -1 = emitted earlier
};

// What the host half leaves for the device half: packets + per-packet plan in one staging buffer,
// the headers of the frames found, the parser state to carry on.  Filling one touches neither a
// context nor the GPU, so the next frame can be planned while the current one is being registered
// (velo_decode_plan_*).
struct DecodePlan {
    bool filled = false;
    size_t n_pkt = 0;
    int nfr = 0, n_lasers = 64, crop_inside = 0;
    bool crop = false, keep_state = false, flush = false;
    double region[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long laser_mask = ~0ull;
    velo_laser_corr corr[64];
    std::vector<velo_pose> carposes;
    std::vector<int64_t> frame_t;
    std::vector<int32_t> frame_packets;
    // staging buffer, owned: [packets | block owners (int16 x 12 per packet) | hdl64 flag per frame |
    // 3x4 table per packet | table-valid flag per packet | median azimuth step per packet], each
    // part 256-byte aligned
    uint8_t* stage = nullptr;
    size_t stage_cap = 0, stage_bytes = 0;
    size_t o_pk = 0, o_blk = 0, o_perm = 0, o_tab = 0, o_tv = 0, o_az = 0;
    void* (*alloc_fn)(size_t) = nullptr;  // staging memory (NULL = malloc / free)
    void (*free_fn)(void*) = nullptr;
    DecodeStream st_next;
    int code = 0;
    char err[200] = {0};
    DecodePlan() = default;
    DecodePlan(const DecodePlan&) = delete;
    DecodePlan& operator=(const DecodePlan&) = delete;
    ~DecodePlan();
    int fail(int c, const char* msg)
    {
        code = c;
        snprintf(err, sizeof err, "%s", msg);
        filled = false;
        return c;
    }
};

// st: the parser state to continue from (a default-constructed one = fresh state, where
// dopts.initial_firing_skip applies).  Returns VELO_OK or a VELO_E_* code (message in P.err).
int decode_plan_host(DecodePlan& P, const DecodeStream& st, const velo_decode_opts& dopts, const uint8_t* packets,
                     const int64_t* pkt_t_us, size_t n_new, const velo_laser_corr corr[64], int n_lasers,
                     const velo_pose* poses, size_t n_poses, int flush, const double* crop_region, int crop_inside,
                     bool keep_state);

}  // namespace velo
