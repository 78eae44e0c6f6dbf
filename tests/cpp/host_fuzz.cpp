// host_fuzz.cpp -- the file parsers and the pose entry points of the host side under
// AddressSanitizer + UBSan (CPU build only: g++ -fsanitize=address,undefined over host/io.cpp,
// host/pose.cpp, host/geodesy.cpp -- no GPU code involved).  Feeds velo_pcap_read / velo_pcap_index /
// velo_carposes_read / velo_load_corrections / velo_insmeta_read truncated, bit-flipped and random
// files, velo_interp_pose / velo_packet_transforms random (also unsorted, duplicated) stores, and the
// host half of the packet decode (host/decode_plan.cpp) random packet streams.
// The only requirement is: no crash, no sanitizer report, sane return codes.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "velo.h"
#include "../../veloslam_amd/csrc/host/decode_plan.hpp"

static void write_file(const std::string& p, const std::vector<uint8_t>& b)
{
    FILE* f = std::fopen(p.c_str(), "wb");
    if (!f) std::exit(3);
    if (!b.empty()) std::fwrite(b.data(), 1, b.size(), f);
    std::fclose(f);
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 300;
    std::mt19937 rng(12345);
    // a valid capture of 40 packets whose azimuth wraps in the middle of packet 17
    std::vector<uint8_t> pk(40 * 1206, 0);
    std::vector<int64_t> ts(40);
    unsigned az = 100;
    for (int i = 0; i < 40; ++i) {
        ts[i] = 1467590400000000LL + 288LL * i;
        for (int b = 0; b < 12; ++b) {
            uint8_t* p = &pk[(size_t)i * 1206 + 100 * b];
            p[0] = 0xff, p[1] = (b & 1) ? 0xdd : 0xee;
            p[2] = az & 0xff, p[3] = az >> 8;
            az = (az + 1700) % 36000;
        }
    }
    const std::string good = dir + "/good.pcap";
    if (velo_pcap_write(good.c_str(), pk.data(), ts.data(), 40)) return 4;
    std::vector<uint8_t> raw;
    {
        FILE* f = std::fopen(good.c_str(), "rb");
        std::fseek(f, 0, SEEK_END);
        raw.resize((size_t)std::ftell(f));
        std::fseek(f, 0, SEEK_SET);
        if (std::fread(raw.data(), 1, raw.size(), f) != raw.size()) return 4;
        std::fclose(f);
    }
    size_t n_ok = 0;
    velo_frame_index idx[64];
    if (velo_pcap_index(good.c_str(), idx, 64, &n_ok) || n_ok < 2) return 5;
    long checked = 0;
    for (int r = 0; r < rounds; ++r) {
        std::vector<uint8_t> b = raw;
        const int mode = r % 4;
        if (mode == 0) b.resize(rng() % (b.size() + 1));                       // truncated anywhere
        else if (mode == 1) for (int k = 0; k < 20; ++k) b[rng() % b.size()] ^= (uint8_t)(1u << (rng() % 8));
        else if (mode == 2) { b.resize(24 + rng() % 4000); for (size_t i = 24; i < b.size(); ++i) b[i] = (uint8_t)rng(); }
        else { for (size_t i = 0; i < b.size() && i < 64; ++i) b[i] = (uint8_t)rng(); }  // header garbage
        const std::string p = dir + "/fuzz.pcap";
        write_file(p, b);
        size_t n = 0, m = 0;
        std::vector<uint8_t> out(64 * 1206);
        std::vector<int64_t> tt(64);
        const int rc1 = velo_pcap_read(p.c_str(), out.data(), tt.data(), 64, &n);
        const int rc2 = velo_pcap_index(p.c_str(), idx, 64, &m);
        if (rc1 > 0 || rc2 > 0 || n > 64 * 4 || (rc2 == 0 && m > 64)) return 6;
        size_t cnt = 0;
        velo_pcap_index(p.c_str(), nullptr, 0, &cnt);
        velo_pose ps[8];
        velo_carposes_read(p.c_str(), ps, 8, &cnt);                           // binary garbage as text
        velo_insmeta_read(p.c_str(), ps, 8, &cnt);
        velo_laser_corr corr[64];
        int32_t en = 0;
        velo_load_corrections(p.c_str(), corr, &en);
        ++checked;
    }
    // carposes: well-formed lines, then junk
    {
        std::string txt;
        for (int i = 0; i < 50; ++i) txt += "1.5 2.5 0.1 0.01 -0.02 10 1467590400 " + std::to_string(10000 * i) + "\n";
        txt += "nan inf -inf 1e400 x y 1 2\n3 4\n";
        write_file(dir + "/c.txt", std::vector<uint8_t>(txt.begin(), txt.end()));
        size_t n = 0;
        std::vector<velo_pose> ps(64);
        if (velo_carposes_read((dir + "/c.txt").c_str(), ps.data(), 64, &n) || n < 50) return 7;
    }
    // pose stores: sorted, with duplicates, unsorted; queries everywhere
    for (int r = 0; r < rounds; ++r) {
        const size_t n = rng() % 40;
        std::vector<velo_pose> st(n);
        int64_t t = 1000000;
        for (size_t i = 0; i < n; ++i) {
            std::memset(&st[i], 0, sizeof(velo_pose));
            const int kind = r % 3;
            t += kind == 2 ? (int64_t)(rng() % 20000) - 10000 : (int64_t)(rng() % (kind ? 3 : 20000));
            st[i].t_us = t;
            st[i].T[0] = (double)i, st[i].R[2] = 0.5 * (double)i, st[i].seconds_pos = 0;
        }
        for (int q = 0; q < 30; ++q) {
            velo_pose out;
            const int64_t tq = 1000000 + (int64_t)(rng() % 900000) - 100000;
            const int rc = velo_interp_pose(st.data(), n, tq, &out);
            if (rc != 0 && rc != VELO_E_NODATA) return 8;
        }
        std::vector<int64_t> pt(20);
        for (auto& v : pt) v = 1000000 + (int64_t)(rng() % 500000);
        std::vector<double> tab(20 * 12);
        std::vector<uint8_t> valid(20);
        velo_pose car;
        if (velo_packet_transforms(st.data(), n, pt.data(), 20, tab.data(), valid.data(), &car)) return 9;
        ++checked;
    }
    // the host half of the decode (host/decode_plan.cpp) on random packet streams: arbitrary
    // azimuths (a split in any block, several per packet), garbage block ids, any number of packets,
    // unsorted stamps, empty / tiny / unsorted pose stores, every option, fresh state and chains of
    // chunked calls carrying the parser state; bad arguments must come back as error codes
    {
        velo_laser_corr corr[64];
        std::memset(corr, 0, sizeof corr);
        velo::DecodePlan P;
        for (int r = 0; r < rounds; ++r) {
            velo::DecodeStream st;
            const int calls = 1 + (int)(rng() % 4);
            for (int c = 0; c < calls; ++c) {
                const size_t n = rng() % 24;
                std::vector<uint8_t> pk2(n * 1206 + 1);
                std::vector<int64_t> t2(n + 1);
                unsigned a2 = rng() % 36000;
                for (size_t i = 0; i < n; ++i) {
                    t2[i] = 1000000 + (int64_t)(rng() % 100000) * (r % 5 == 0 ? 1 : (int64_t)i);
                    for (size_t k = 0; k < 1206; ++k) pk2[i * 1206 + k] = (uint8_t)rng();
                    if (r % 3) {  // mostly plausible azimuths, sometimes pure noise
                        for (int b = 0; b < 12; ++b) {
                            uint8_t* q = &pk2[i * 1206 + 100 * (size_t)b];
                            q[0] = 0xff, q[1] = (rng() & 1) ? 0xdd : 0xee;
                            q[2] = a2 & 0xff, q[3] = (uint8_t)(a2 >> 8);
                            a2 = (a2 + 20 + rng() % 6000) % 36000;
                        }
                    }
                }
                const size_t np = r % 4 == 0 ? 0 : rng() % 6;
                std::vector<velo_pose> ps2(np + 1);
                for (size_t i = 0; i < np; ++i) {
                    std::memset(&ps2[i], 0, sizeof(velo_pose));
                    ps2[i].t_us = 1000000 + (int64_t)(rng() % 2000000);
                    ps2[i].T[0] = (double)(rng() % 100), ps2[i].R[2] = (double)(rng() % 360);
                    ps2[i].seconds_pos = (rng() % 5) ? 0.0 : -1.0;
                }
                velo_decode_opts o;
                std::memset(&o, 0, sizeof o);
                o.struct_size = sizeof o;
                o.points_skip = (int)(rng() % 4);
                o.initial_firing_skip = (int)(rng() % 13);
                for (auto& l : o.laser_selection) l = (uint8_t)(rng() & 1);
                const int lasers[4] = {64, 32, 16, 7};
                const int nl = lasers[rng() % 4];
                const double crop[6] = {-1, 1, -2, 2, -3, 3};
                const int rc = velo::decode_plan_host(P, st, o, n ? pk2.data() : nullptr, n ? t2.data() : nullptr, n,
                                                      (rng() % 16) ? corr : nullptr, nl, np ? ps2.data() : nullptr, np,
                                                      (int)(rng() & 1), (rng() & 1) ? crop : nullptr, (int)(rng() & 1), true);
                if (rc > 0) return 10;
                if (rc == VELO_OK) {
                    if (!P.filled || P.nfr < 0 || (size_t)P.nfr != P.carposes.size() || P.stage_bytes > P.stage_cap) return 11;
                    const int16_t* blk = reinterpret_cast<const int16_t*>(P.stage + P.o_blk);
                    for (size_t i = 0; i < P.n_pkt * 12; ++i)
                        if (blk[i] < -1 || blk[i] >= P.nfr) return 12;   // an owner is an emitted frame or nobody
                    st = P.flush ? velo::DecodeStream() : std::move(P.st_next);
                }
                ++checked;
            }
        }
    }
    std::printf("host fuzz: %ld cases, frames in the good capture: %zu\n", checked, n_ok);
    return 0;
}
