// host_fuzz.cpp -- the file parsers and the pose entry points of the host side under
// AddressSanitizer + UBSan (CPU build only: g++ -fsanitize=address,undefined over host/io.cpp,
// host/pose.cpp, host/geodesy.cpp -- no GPU code involved).  Feeds velo_pcap_read / velo_pcap_index /
// velo_carposes_read / velo_load_corrections / velo_insmeta_read truncated, bit-flipped and random
// files, and velo_interp_pose / velo_packet_transforms random (also unsorted, duplicated) stores.
// The only requirement is: no crash, no sanitizer report, sane return codes.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "velo.h"

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
    std::printf("host fuzz: %ld cases, frames in the good capture: %zu\n", checked, n_ok);
    return 0;
}
