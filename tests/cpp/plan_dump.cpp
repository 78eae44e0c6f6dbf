// plan_dump.cpp -- the HOST half of the decode (veloslam_amd/csrc/host/decode_plan.cpp) on its own,
// no GPU, no library: g++ over this file + host/decode_plan.cpp + host/pose.cpp + host/geodesy.cpp.
// Reads a packet stream written by tests/test_host_parity.py, runs the sequential part of the parser
// one-shot or in chunks (carrying the parser state exactly as velo_decode_stream does) and prints what
// it found per frame -- for the test to hold against the oracle's parser (oracle/decode.c).
//   plan_dump DIR first_block points_skip chunk [chunk ...]     (chunk sizes in packets; 0 = all)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "../../veloslam_amd/csrc/host/decode_plan.hpp"

template <typename T>
static std::vector<T> slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", path.c_str());
        std::exit(2);
    }
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<T> v((size_t)n / sizeof(T));
    f.read(reinterpret_cast<char*>(v.data()), n);
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const std::string dir = argv[1];
    const auto pk = slurp<uint8_t>(dir + "/packets.bin");
    const auto ts = slurp<int64_t>(dir + "/times.i64");
    const auto poses = slurp<velo_pose>(dir + "/poses.bin");
    const auto corr = slurp<velo_laser_corr>(dir + "/corr.bin");
    if (corr.size() != 64 || pk.size() != ts.size() * 1206) return 3;
    velo_decode_opts o;
    std::memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    o.initial_firing_skip = std::atoi(argv[2]);
    o.points_skip = std::atoi(argv[3]);
    std::memset(o.laser_selection, 1, sizeof o.laser_selection);
    velo::DecodePlan P;
    velo::DecodeStream st;
    size_t at = 0;
    int frame = 0;
    long owned_total = 0;
    for (int a = 4; a < argc && at < ts.size(); ++a) {
        size_t n = (size_t)std::atol(argv[a]);
        if (n == 0 || at + n > ts.size()) n = ts.size() - at;
        const bool last = at + n == ts.size();
        if (velo::decode_plan_host(P, st, o, pk.data() + at * 1206, ts.data() + at, n, corr.data(), 64, poses.data(),
                                   poses.size(), last ? 1 : 0, nullptr, 0, true)) {
            std::fprintf(stderr, "plan failed: %s\n", P.err);
            return 4;
        }
        at += n;
        // blocks owned by the frames this call emits, per frame, out of the staged owner table
        std::vector<long> owned((size_t)P.nfr, 0);
        const int16_t* blk = reinterpret_cast<const int16_t*>(P.stage + P.o_blk);
        for (size_t i = 0; i < P.n_pkt * 12; ++i)
            if (blk[i] >= 0 && blk[i] < P.nfr) ++owned[(size_t)blk[i]];
        const uint8_t* tv = P.stage + P.o_tv;
        long valid = 0;
        for (size_t i = 0; i < P.n_pkt; ++i) valid += tv[i];
        for (int f = 0; f < P.nfr; ++f, ++frame) {
            const velo_pose& c = P.carposes[(size_t)f];
            std::printf("frame %d t %lld packets %d blocks %ld car %.17g %.17g %.17g %.17g %.17g %.17g sp %g\n", frame,
                        (long long)P.frame_t[(size_t)f], P.frame_packets[(size_t)f], owned[(size_t)f], c.T[0], c.T[1],
                        c.T[2], c.R[0], c.R[1], c.R[2], c.seconds_pos);
            owned_total += owned[(size_t)f];
        }
        std::printf("call packets_in_flight %zu tables_valid %ld emitted %d\n", P.n_pkt, valid, P.nfr);
        st = last ? velo::DecodeStream() : std::move(P.st_next);   // what decode_submit does with keep_state
    }
    std::printf("total frames %d blocks %ld\n", frame, owned_total);
    return 0;
}
