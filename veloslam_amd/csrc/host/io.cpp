// io.cpp -- the data formats on either side of the path, host-only:
//   * pcap files of Velodyne packets without libpcap (vtkPacketFileReader.h:57-66,166-197;
//     vtkPacketFileWriter.cxx:41-54,118-161): 24-byte global header, 16-byte record header,
//     42-byte Ethernet/IPv4/UDP prefix, 1206-byte payload -> 1264 bytes per packet on disk;
//   * the INS sample -> pose step (INSSource.cxx:305-326, InsPVA wire struct
//     type_defs.h:39-58) and the pose-store file formats (carposes.txt is in pose.cpp;
//     .insmeta record = type_defs.cxx:4-33 with ptime flattened to int64 microseconds).
#include <cmath>
#include <cstdlib>
#include <iterator>
#include <string>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>
#include "../../../include/veloslam/TransformManager.hpp"

namespace {

// Ethernet + IPv4 + UDP prefix the reference writer puts in front of every lidar packet,
// built field by field: broadcast dst, Velodyne-OUI src, 192.168.1.200:2368 -> broadcast:2368.
void lidar_prefix(uint8_t h[42])
{
    std::memset(h, 0, 42);
    for (int i = 0; i < 6; ++i) h[i] = 0xff;  // dst MAC
    h[6] = 0x60; h[7] = 0x76; h[8] = 0x88;     // src MAC 60:76:88:00:00:00
    h[12] = 0x08; h[13] = 0x00;                // IPv4
    h[14] = 0x45;                              // version 4, IHL 5
    h[16] = 0x04; h[17] = 0xd2;                // total length 1234
    h[20] = 0x40;                              // don't fragment
    h[22] = 0xff; h[23] = 0x11;                // TTL 255, UDP
    h[24] = 0xb4; h[25] = 0xaa;                // header checksum as written by the reference
    h[26] = 0xc0; h[27] = 0xa8; h[28] = 0x01; h[29] = 0xc8;  // 192.168.1.200
    h[30] = h[31] = h[32] = h[33] = 0xff;      // 255.255.255.255
    h[34] = 0x09; h[35] = 0x40;                // src port 2368
    h[36] = 0x09; h[37] = 0x40;                // dst port 2368
    h[38] = 0x04; h[39] = 0xbe;                // UDP length 1214
}

void put32(std::vector<uint8_t>& b, uint32_t v)
{
    for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i)));
}
uint32_t get32(const uint8_t* p, bool swap)
{
    return swap ? ((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3])
                : ((uint32_t)p[3] << 24 | (uint32_t)p[2] << 16 | (uint32_t)p[1] << 8 | p[0]);
}

}  // namespace

// ---- tag scanner for the calibration file (velo_load_corrections below)
namespace {

// text of the first <name ...>...</name> inside [from, to); empty if absent
bool tag_text(const std::string& s, size_t from, size_t to, const char* name, std::string* out,
              size_t* after = nullptr)
{
    const std::string open = std::string("<") + name;
    size_t a = from;
    for (;;) {
        a = s.find(open, a);
        if (a == std::string::npos || a >= to) return false;
        const char c = a + open.size() < s.size() ? s[a + open.size()] : '\0';
        if (c == '>' || c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '/') break;
        a += open.size();  // a longer tag name that merely starts the same
    }
    const size_t gt = s.find('>', a);
    if (gt == std::string::npos || gt >= to) return false;
    if (s[gt - 1] == '/') {  // <name/>
        out->clear();
        if (after) *after = gt + 1;
        return true;
    }
    const std::string close = std::string("</") + name + ">";
    const size_t b = s.find(close, gt + 1);
    if (b == std::string::npos || b > to) return false;
    *out = s.substr(gt + 1, b - gt - 1);
    if (after) *after = b + close.size();
    return true;
}

// [start of content, start of closing tag) of the first <name> element inside [from, to)
bool tag_span(const std::string& s, size_t from, size_t to, const char* name, size_t* a, size_t* b)
{
    const std::string open = std::string("<") + name;
    size_t p = from;
    for (;;) {
        p = s.find(open, p);
        if (p == std::string::npos || p >= to) return false;
        const char c = p + open.size() < s.size() ? s[p + open.size()] : '\0';
        if (c == '>' || c == ' ' || c == '\t' || c == '\n' || c == '\r') break;
        p += open.size();
    }
    const size_t gt = s.find('>', p);
    const std::string close = std::string("</") + name + ">";
    const size_t e = gt == std::string::npos ? std::string::npos : s.find(close, gt + 1);
    if (e == std::string::npos || e > to) return false;
    *a = gt + 1;
    *b = e;
    return true;
}

}  // namespace

extern "C" {

int velo_pcap_write(const char* path, const uint8_t* packets, const int64_t* t_us, size_t n_pkt)
{
    if (!path || (n_pkt && (!packets || !t_us))) return VELO_E_INVALID;
    std::vector<uint8_t> buf;
    buf.reserve(24 + n_pkt * 1264);
    put32(buf, 0xa1b2c3d4u);  // microsecond pcap, little endian
    buf.push_back(2); buf.push_back(0); buf.push_back(4); buf.push_back(0);  // version 2.4
    put32(buf, 0);            // thiszone
    put32(buf, 0);            // sigfigs
    put32(buf, 65535);        // snaplen
    put32(buf, 1);            // LINKTYPE_ETHERNET
    uint8_t prefix[42];
    lidar_prefix(prefix);
    for (size_t i = 0; i < n_pkt; ++i) {
        put32(buf, (uint32_t)(t_us[i] / 1000000));
        put32(buf, (uint32_t)(t_us[i] % 1000000));
        put32(buf, 1248);
        put32(buf, 1248);
        buf.insert(buf.end(), prefix, prefix + 42);
        buf.insert(buf.end(), packets + i * 1206, packets + (i + 1) * 1206);
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) return VELO_E_NODATA;
    const size_t w = std::fwrite(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    return w == buf.size() ? VELO_OK : VELO_E_NODATA;
}

// Reads every UDP datagram whose payload is 1206 bytes (what HDLParser accepts,
// HDLParser.cxx:982-985).  packets/t_us may be NULL to count; cap = capacity in packets.
int velo_pcap_read(const char* path, uint8_t* packets, int64_t* t_us, size_t cap, size_t* n_out)
{
    if (!path || !n_out) return VELO_E_INVALID;
    *n_out = 0;
    FILE* f = std::fopen(path, "rb");
    if (!f) return VELO_E_NODATA;
    uint8_t gh[24];
    if (std::fread(gh, 1, 24, f) != 24) {
        std::fclose(f);
        return VELO_E_INVALID;
    }
    const uint32_t magic = get32(gh, false);
    bool swap = false, nano = false;
    if (magic == 0xa1b2c3d4u) {
    } else if (magic == 0xa1b23c4du) {
        nano = true;
    } else if (magic == 0xd4c3b2a1u) {
        swap = true;
    } else if (magic == 0x4d3cb2a1u) {
        swap = nano = true;
    } else {
        std::fclose(f);
        return VELO_E_INVALID;
    }
    std::vector<uint8_t> rec;
    size_t n = 0;
    for (;;) {
        uint8_t rh[16];
        if (std::fread(rh, 1, 16, f) != 16) break;
        const uint32_t sec = get32(rh, swap), frac = get32(rh + 4, swap), incl = get32(rh + 8, swap);
        if (incl > (1u << 20)) break;  // corrupt
        rec.resize(incl);
        if (incl && std::fread(rec.data(), 1, incl, f) != incl) break;
        // Ethernet II + IPv4 (no options) + UDP, payload 1206
        if (incl != 1248 || rec[12] != 0x08 || rec[13] != 0x00 || (rec[14] & 0x0f) != 5 || rec[23] != 17)
            continue;
        if (packets && t_us) {
            if (n >= cap) {
                std::fclose(f);
                *n_out = n;
                return VELO_E_RANGE;
            }
            std::memcpy(packets + n * 1206, rec.data() + 42, 1206);
            t_us[n] = (int64_t)sec * 1000000 + (nano ? frac / 1000 : frac);
        }
        ++n;
    }
    std::fclose(f);
    *n_out = n;
    return VELO_OK;
}

// HDLParser::readFrameInformation (HDLParser.cxx:1065-1160): one pass over the capture that
// looks at nothing but the rotational position of every firing block.  The first frame starts at
// the first record (file position right after the 24-byte global header, skip 0, time of the first
// lidar packet); a frame boundary is a block whose RAW azimuth (no modulo, HDLParser.cxx:1126) is
// below the previous block's, and the new frame is { position of the record that holds that
// block, index of the block, time of that packet } -- exactly what HDLManager::loadOffline
// (HDLManager.cxx:103-117) stores per frame and HDLParser::getFrame (HDLParser.cxx:505-544) seeks
// to.  Like the reference, the position remembered is the one after the last 1206-byte packet,
// so foreign records in between are skipped again on re-read.
int velo_pcap_index(const char* path, velo_frame_index* frames, size_t cap, size_t* n_out)
{
    if (!path || !n_out) return VELO_E_INVALID;
    *n_out = 0;
    FILE* f = std::fopen(path, "rb");
    if (!f) return VELO_E_NODATA;
    uint8_t gh[24];
    if (std::fread(gh, 1, 24, f) != 24) {
        std::fclose(f);
        return VELO_E_INVALID;
    }
    const uint32_t magic = get32(gh, false);
    bool swap = false, nano = false;
    if (magic == 0xa1b2c3d4u) {
    } else if (magic == 0xa1b23c4du) {
        nano = true;
    } else if (magic == 0xd4c3b2a1u) {
        swap = true;
    } else if (magic == 0x4d3cb2a1u) {
        swap = nano = true;
    } else {
        std::fclose(f);
        return VELO_E_INVALID;
    }
    size_t n = 0;
    bool overflow = false;
    auto push = [&](int64_t pos, int32_t skip, int64_t pkt, int64_t t) {
        if (frames && n >= cap) overflow = true;  // keep counting: *n_out = frames found
        if (frames && n < cap) {
            frames[n].file_pos = pos;
            frames[n].firing_skip = skip;
            frames[n].reserved = 0;
            frames[n].first_packet = pkt;
            frames[n].t_us = t;
        }
        ++n;
    };
    int64_t last_pos = 24, n_pkt = 0;
    unsigned last_az = 0;
    bool have_first = false;
    push(last_pos, 0, 0, VELO_TIME_INVALID);
    std::vector<uint8_t> rec;
    for (;;) {
        uint8_t rh[16];
        if (std::fread(rh, 1, 16, f) != 16) break;
        const uint32_t sec = get32(rh, swap), frac = get32(rh + 4, swap), incl = get32(rh + 8, swap);
        if (incl > (1u << 20)) break;
        rec.resize(incl);
        if (incl && std::fread(rec.data(), 1, incl, f) != incl) break;
        if (incl != 1248 || rec[12] != 0x08 || rec[13] != 0x00 || (rec[14] & 0x0f) != 5 || rec[23] != 17)
            continue;  // not a 1206-byte lidar datagram: position NOT remembered, as in the reference
        const int64_t t = (int64_t)sec * 1000000 + (nano ? frac / 1000 : frac);
        if (!have_first) {
            have_first = true;
            if (frames && cap) frames[0].t_us = t;
        }
        const uint8_t* pk = rec.data() + 42;
        for (int b = 0; b < 12; ++b) {
            const unsigned az = (unsigned)pk[b * 100 + 2] | ((unsigned)pk[b * 100 + 3] << 8);
            if (az < last_az) push(last_pos, b, n_pkt, t);
            last_az = az;
        }
        ++n_pkt;
        last_pos = (int64_t)std::ftell(f);
    }
    std::fclose(f);
    *n_out = n;
    return overflow ? VELO_E_RANGE : VELO_OK;
}

// INSSource::calcTransform (INSSource.cxx:305-326): LLH degrees -> radians -> ENU about
// orig_xyz (the reference's default is {-2781621.9891904, 4672106.75052387, 18.8910392},
// INSSource.cxx:334 -- passed in, never "fixed"); Euler angles and velocity copied.  The
// time stamp is the caller's (TimeSolver is out of scope).
int velo_ins_to_pose(const velo_inspva* ins, const double orig_xyz[3], int64_t t_us, velo_pose* out)
{
    if (!ins || !orig_xyz || !out) return VELO_E_INVALID;
    double in[3] = {ins->LLH[0] * M_PI / 180, ins->LLH[1] * M_PI / 180, ins->LLH[2]};
    double org[3] = {orig_xyz[0], orig_xyz[1], orig_xyz[2]};
    double enu[3] = {0, 0, 0};
    llh2enu(in, org, enu);
    std::memset(out, 0, sizeof *out);
    for (int i = 0; i < 3; ++i) {
        out->T[i] = enu[i];
        out->R[i] = ins->Eulr[i];
        out->V[i] = ins->V[i];
    }
    out->week_number = ins->week_number;
    out->milliseconds = ins->milliseconds;
    out->week_number_pos = ins->week_number_pos;
    out->seconds_pos = ins->seconds_pos;
    out->t_us = t_us;
    return VELO_OK;
}

// .insmeta: the reference dumps T[i],R[i],V[i] interleaved, then timestamp, week_number,
// milliseconds, week_number_pos, seconds_pos (type_defs.cxx:4-33); the timestamp is a
// boost::ptime there (ABI-specific bytes) and int64 microseconds here.  98 bytes per record.
int velo_insmeta_write(const char* path, const velo_pose* poses, size_t n)
{
    if (!path || (n && !poses)) return VELO_E_INVALID;
    std::ofstream os(path, std::ios::binary);
    if (!os) return VELO_E_NODATA;
    for (size_t k = 0; k < n; ++k) {
        const velo_pose& p = poses[k];
        for (int i = 0; i < 3; ++i) {
            os.write(reinterpret_cast<const char*>(&p.T[i]), 8);
            os.write(reinterpret_cast<const char*>(&p.R[i]), 8);
            os.write(reinterpret_cast<const char*>(&p.V[i]), 8);
        }
        os.write(reinterpret_cast<const char*>(&p.t_us), 8);
        os.write(reinterpret_cast<const char*>(&p.week_number), 2);
        os.write(reinterpret_cast<const char*>(&p.milliseconds), 4);
        os.write(reinterpret_cast<const char*>(&p.week_number_pos), 4);
        os.write(reinterpret_cast<const char*>(&p.seconds_pos), 8);
    }
    return os ? VELO_OK : VELO_E_NODATA;
}

int velo_insmeta_read(const char* path, velo_pose* poses, size_t cap, size_t* n_out)
{
    if (!path || !n_out) return VELO_E_INVALID;
    *n_out = 0;
    std::ifstream is(path, std::ios::binary);
    if (!is) return VELO_E_NODATA;
    size_t n = 0;
    for (;;) {
        velo_pose p;
        std::memset(&p, 0, sizeof p);
        for (int i = 0; i < 3; ++i) {
            is.read(reinterpret_cast<char*>(&p.T[i]), 8);
            is.read(reinterpret_cast<char*>(&p.R[i]), 8);
            is.read(reinterpret_cast<char*>(&p.V[i]), 8);
        }
        is.read(reinterpret_cast<char*>(&p.t_us), 8);
        is.read(reinterpret_cast<char*>(&p.week_number), 2);
        is.read(reinterpret_cast<char*>(&p.milliseconds), 4);
        is.read(reinterpret_cast<char*>(&p.week_number_pos), 4);
        is.read(reinterpret_cast<char*>(&p.seconds_pos), 8);
        if (!is) break;
        if (poses) {
            if (n >= cap) {
                *n_out = n;
                return VELO_E_RANGE;
            }
            poses[n] = p;
        }
        ++n;
    }
    *n_out = n;
    return VELO_OK;
}

// ---- calibration file (HDLParser::vsInternal::loadCorrectionsFile, HDLParser.cxx:771-858) ----
// The reference reads Velodyne's db.xml through boost::property_tree; only the element names
// matter to it, so a plain tag scanner is enough: boost_serialization.DB.enabled_ (items equal
// to 1 are counted) and boost_serialization.DB.points_ (item > px > id_, rotCorrection_,
// vertCorrection_, distCorrection_, vertOffsetCorrection_, horizOffsetCorrection_).  Units and
// derived fields as there: the three distances are centimetres in the file and metres
// afterwards (:836-838), sin/cos of the vertical angle (:840-841), the two offset products
// (:848-855).  Lasers the file does not mention stay zero.
int velo_load_corrections(const char* path, velo_laser_corr corr[64], int32_t* n_enabled)
{
    if (!path || !corr) return VELO_E_INVALID;
    std::ifstream is(path, std::ios::binary);
    if (!is) return VELO_E_NODATA;
    const std::string s((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
    std::memset(corr, 0, 64 * sizeof(velo_laser_corr));
    size_t da, db;
    if (!tag_span(s, 0, s.size(), "DB", &da, &db)) return VELO_E_NODATA;
    int enabled = 0;
    size_t ea, eb;
    if (tag_span(s, da, db, "enabled_", &ea, &eb)) {
        size_t p = ea;
        std::string t;
        while (tag_text(s, p, eb, "item", &t, &p)) {
            char* end = nullptr;
            const long v = std::strtol(t.c_str(), &end, 10);
            if (end != t.c_str() && v == 1) ++enabled;
        }
    }
    if (n_enabled) *n_enabled = enabled;
    size_t pa, pb;
    if (!tag_span(s, da, db, "points_", &pa, &pb)) return VELO_E_NODATA;
    size_t p = pa;
    for (;;) {
        size_t xa, xb;
        if (!tag_span(s, p, pb, "px", &xa, &xb)) break;
        p = xb + 5;  // past </px>
        std::string t;
        int index = -1;
        double az = 0, vert = 0, dist = 0, voff = 0, hoff = 0;
        if (tag_text(s, xa, xb, "id_", &t)) index = std::atoi(t.c_str());
        if (tag_text(s, xa, xb, "rotCorrection_", &t)) az = std::atof(t.c_str());
        if (tag_text(s, xa, xb, "vertCorrection_", &t)) vert = std::atof(t.c_str());
        if (tag_text(s, xa, xb, "distCorrection_", &t)) dist = std::atof(t.c_str());
        if (tag_text(s, xa, xb, "vertOffsetCorrection_", &t)) voff = std::atof(t.c_str());
        if (tag_text(s, xa, xb, "horizOffsetCorrection_", &t)) hoff = std::atof(t.c_str());
        if (index < 0 || index >= 64) continue;  // (the reference would index out of bounds)
        velo_laser_corr& c = corr[index];
        c.azimuthCorrection = az;
        c.verticalCorrection = vert;
        c.distanceCorrection = dist / 100.0;
        c.verticalOffsetCorrection = voff / 100.0;
        c.horizontalOffsetCorrection = hoff / 100.0;
        c.cosVertCorrection = std::cos(c.verticalCorrection * M_PI / 180.0);
        c.sinVertCorrection = std::sin(c.verticalCorrection * M_PI / 180.0);
    }
    for (int i = 0; i < 64; ++i) {
        corr[i].sinVertOffsetCorrection = corr[i].verticalOffsetCorrection * corr[i].sinVertCorrection;
        corr[i].cosVertOffsetCorrection = corr[i].verticalOffsetCorrection * corr[i].cosVertCorrection;
    }
    return VELO_OK;
}

}  // extern "C"
