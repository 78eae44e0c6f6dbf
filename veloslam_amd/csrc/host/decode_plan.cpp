// decode_plan.cpp -- see decode_plan.hpp.  The loop below follows HDLParser::processHDLPacket
// (HDLParser.cxx:980-1055) statement by statement (line references inline); oracle/decode.c is the
// restatement it is held to.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "decode_plan.hpp"
#include "../../../include/veloslam/TransformManager.hpp"

namespace velo {

DecodePlan::~DecodePlan()
{
    if (stage) (free_fn ? free_fn : std::free)(stage);
}

int decode_plan_host(DecodePlan& P, const DecodeStream& st, const velo_decode_opts& dopts, const uint8_t* packets,
                     const int64_t* pkt_t_us, size_t n_new, const velo_laser_corr corr[64], int n_lasers,
                     const velo_pose* poses, size_t n_poses, int flush, const double* crop_region, int crop_inside,
                     bool keep_state)
{
    P.filled = false;
    P.code = 0;
    if ((n_new && (!packets || !pkt_t_us)) || !corr || (n_poses && !poses))
        return P.fail(VELO_E_INVALID, "velo_decode: null argument");
    if (n_lasers != 64 && n_lasers != 32 && n_lasers != 16)
        return P.fail(VELO_E_INVALID, "n_lasers must be 64, 32 or 16");
    const size_t n_pend = st.t.size();
    const size_t n_pkt = n_pend + n_new;
    if (n_pkt == 0 || n_pkt > 60000) return P.fail(VELO_E_RANGE, "packets in flight must be in [1, 60000]");

    const veloslam::SortedPoseView tm(poses, n_poses);  // the caller's store, read in place: O(log n) per packet
    // working set = what the unfinished frame still needs + the new packets
    // (a parse that starts from fresh state -- every call of velo_decode -- reads the caller's
    // packets in place: no 360 KB copy per frame in front of the copy into the pinned stage)
    std::vector<uint8_t> bytes_joined;
    if (!st.bytes.empty()) {
        bytes_joined = st.bytes;
        bytes_joined.insert(bytes_joined.end(), packets, packets + n_new * 1206);
    }
    const uint8_t* const bytes_p = st.bytes.empty() ? packets : bytes_joined.data();
    std::vector<int64_t> times(st.t);
    times.insert(times.end(), pkt_t_us, pkt_t_us + n_new);
    std::vector<int16_t> blk(st.blk);
    blk.resize(n_pkt * 12, -1);
    std::vector<double> table(st.table);
    table.resize(n_pkt * 12, 0.0);
    std::vector<uint8_t> tvalid(st.tvalid), perm;
    tvalid.resize(n_pkt, 0);
    std::vector<int32_t> azdiff(st.azdiff);
    azdiff.resize(n_pkt, 0);
    P.carposes.clear();
    P.frame_t.clear();
    P.frame_packets.clear();
    // a parse that starts from fresh state begins at the configured block (getFrame's `skip`)
    const bool fresh = !st.inited && !st.open && st.t.empty() && st.last_az == -1;
    int last_az = st.last_az, firing_skip = fresh ? dopts.initial_firing_skip : st.firing_skip, cur = 0;
    const int pskip = dopts.points_skip;
    bool inited = st.inited, is_hdl64 = st.is_hdl64;
    veloslam::PoseTransform carpose = st.carpose;
    auto open_frame = [&]() {
        P.carposes.push_back(veloslam::PoseTransform().toC());
        P.frame_t.push_back(VELO_TIME_INVALID);
        P.frame_packets.push_back(0);
        perm.push_back(0);
    };
    open_frame();
    if (st.open) {  // header of the frame the previous call left unfinished
        P.carposes[0] = st.carpose0;
        P.frame_t[0] = st.frame_t;
        P.frame_packets[0] = st.frame_packets;
    }
    for (size_t p = n_pend; p < n_pkt; ++p) {
        const uint8_t* d = bytes_p + p * 1206;
        veloslam::PoseTransform tr;
        tm.interpolate(times[p], &tr);
        if (!inited) {  // :992-1001
            carpose = tr;
            P.carposes[cur] = tr.toC();
            P.frame_t[cur] = times[p];
            P.frame_packets[cur]++;
            inited = true;
        }
        tr.timestamp = times[p];
        if (tr.seconds_pos != -1) {  // :1004-1007 (+ :1057-1062)
            for (int a = 0; a < 3; ++a) tr.T[a] -= carpose.T[a];
            const veloslam::Affine3x4 M = tr.getMatrix();
            std::memcpy(&table[p * 12], M.data(), 12 * sizeof(double));
            tvalid[p] = 1;
        }
        P.frame_packets[cur]++;  // :1009
        int block = firing_skip;
        firing_skip = 0;
        int diffs[11];
        for (int i = 0; i < 11; ++i) {
            const int r1 = d[100 * (i + 1) + 2] | (d[100 * (i + 1) + 3] << 8);
            const int r0 = d[100 * i + 2] | (d[100 * i + 3] << 8);
            diffs[i] = (36000 + r1 - r0) % 36000;
        }
        std::sort(diffs, diffs + 11);
        azdiff[p] = diffs[6];  // nth_element(..., 12/2): element 6 of 11, :1021-1026
        for (; block < 12; ++block) {
            const uint8_t* fd = d + 100 * block;
            const unsigned id = fd[0] | (fd[1] << 8);
            const int rot = fd[2] | (fd[3] << 8);
            is_hdl64 |= (id != 0xeeff);
            if (rot < last_az) {  // :1035-1039 -> splitFrame
                firing_skip = block;
                perm[cur] = is_hdl64 ? 1 : 0;
                ++cur;
                if (cur >= 32000) return P.fail(VELO_E_RANGE, "too many frames in one decode call");
                open_frame();
                inited = false;
            }
            // :1042 -- a skipped block still takes part in the split logic above
            if (pskip == 0 || block % (pskip + 1) == 0) blk[p * 12 + block] = (int16_t)cur;
            last_az = rot;
        }
    }
    int nfr = cur;
    P.st_next = DecodeStream();
    if (keep_state && !flush) {
        // carry the parser on: keep the packets that hold blocks of the unfinished frame `cur`
        DecodeStream nx;
        nx.last_az = last_az;
        nx.firing_skip = firing_skip;
        nx.inited = inited;
        nx.is_hdl64 = is_hdl64;
        nx.carpose = carpose;
        nx.open = true;
        nx.carpose0 = P.carposes[cur];
        nx.frame_t = P.frame_t[cur];
        nx.frame_packets = P.frame_packets[cur];
        size_t p0 = n_pkt;
        for (size_t p = 0; p < n_pkt && p0 == n_pkt; ++p)
            for (int k = 0; k < 12; ++k)
                if (blk[p * 12 + k] == cur) {
                    p0 = p;
                    break;
                }
        for (size_t p = p0; p < n_pkt; ++p) {
            nx.bytes.insert(nx.bytes.end(), bytes_p + p * 1206, bytes_p + (p + 1) * 1206);
            nx.t.push_back(times[p]);
            nx.table.insert(nx.table.end(), table.begin() + p * 12, table.begin() + (p + 1) * 12);
            nx.tvalid.push_back(tvalid[p]);
            nx.azdiff.push_back(azdiff[p]);
            for (int k = 0; k < 12; ++k) nx.blk.push_back(blk[p * 12 + k] == cur ? (int16_t)0 : (int16_t)-1);
        }
        P.st_next = std::move(nx);
    }
    if (flush) {
        perm[cur] = is_hdl64 ? 1 : 0;
        nfr = cur + 1;
    } else {
        for (auto& b : blk)
            if (b == cur) b = -1;  // the unfinished frame is not emitted
        P.carposes.resize((size_t)nfr);
        P.frame_t.resize((size_t)nfr);
        P.frame_packets.resize((size_t)nfr);
    }
    // Packets and per-packet plan go up in ONE copy from a pinned staging buffer (six copies from
    // pageable vectors each blocked the host for their staging).
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    P.o_pk = 0;
    P.o_blk = al(P.o_pk + n_pkt * 1206);
    P.o_perm = al(P.o_blk + blk.size() * sizeof(int16_t));
    P.o_tab = al(P.o_perm + perm.size());
    P.o_tv = al(P.o_tab + table.size() * sizeof(double));
    P.o_az = al(P.o_tv + n_pkt);
    P.stage_bytes = al(P.o_az + n_pkt * sizeof(int32_t));
    if (P.stage_cap < P.stage_bytes) {
        if (P.stage) (P.free_fn ? P.free_fn : std::free)(P.stage);
        P.stage = nullptr;
        P.stage_cap = 0;
        const size_t want = P.stage_bytes + P.stage_bytes / 2;
        P.stage = static_cast<uint8_t*>(P.alloc_fn ? P.alloc_fn(want) : std::malloc(want));
        if (!P.stage) return P.fail(VELO_E_NOMEM, "velo_decode: no memory for the staging buffer");
        P.stage_cap = want;
    }
    std::memcpy(P.stage + P.o_pk, bytes_p, n_pkt * 1206);
    std::memcpy(P.stage + P.o_blk, blk.data(), blk.size() * sizeof(int16_t));
    std::memcpy(P.stage + P.o_perm, perm.data(), perm.size());
    std::memcpy(P.stage + P.o_tab, table.data(), table.size() * sizeof(double));
    std::memcpy(P.stage + P.o_tv, tvalid.data(), n_pkt);
    std::memcpy(P.stage + P.o_az, azdiff.data(), n_pkt * sizeof(int32_t));
    P.n_pkt = n_pkt;
    P.nfr = nfr;
    P.n_lasers = n_lasers;
    std::memcpy(P.corr, corr, sizeof P.corr);
    P.crop = crop_region != nullptr;
    P.crop_inside = crop_inside;
    for (int i = 0; i < 6; ++i) P.region[i] = crop_region ? crop_region[i] : 0.0;
    P.laser_mask = 0;
    for (int i = 0; i < 64; ++i)
        if (dopts.laser_selection[i]) P.laser_mask |= 1ull << i;
    P.keep_state = keep_state;
    P.flush = flush != 0;
    P.filled = true;
    return VELO_OK;
}

}  // namespace velo
