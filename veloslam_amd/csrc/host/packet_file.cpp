// packet_file.cpp -- veloslam::PacketFileReader / PacketFileWriter (include/veloslam/PacketFile.hpp):
// the record-by-record forms of the capture container host/io.cpp reads and writes in bulk.
#include <chrono>
#include <cstring>
#include "../../../include/veloslam/PacketFile.hpp"

namespace veloslam {

namespace {
uint32_t rd32(const unsigned char* p, bool swap)
{
    return swap ? ((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3])
                : ((uint32_t)p[3] << 24 | (uint32_t)p[2] << 16 | (uint32_t)p[1] << 8 | p[0]);
}
void wr32(unsigned char* p, uint32_t v)
{
    for (int i = 0; i < 4; ++i) p[i] = (unsigned char)(v >> (8 * i));
}
// Ethernet II + IPv4 + UDP in front of a payload, field by field: broadcast destination, Velodyne
// source (60:76:88:00:00:00, 192.168.1.200), the port on both sides, the UDP length; the IPv4
// total length and checksum are the lidar packet's in BOTH prefixes, as the reference writes them
// (vtkPacketFileWriter.cxx:36-54)
void prefix(unsigned char h[42], unsigned port, unsigned udp_len)
{
    std::memset(h, 0, 42);
    for (int i = 0; i < 6; ++i) h[i] = 0xff;
    h[6] = 0x60, h[7] = 0x76, h[8] = 0x88;
    h[12] = 0x08, h[13] = 0x00;
    h[14] = 0x45;
    h[16] = 0x04, h[17] = 0xd2;
    h[20] = 0x40;
    h[22] = 0xff, h[23] = 0x11;
    h[24] = 0xb4, h[25] = 0xaa;
    h[26] = 0xc0, h[27] = 0xa8, h[28] = 0x01, h[29] = 0xc8;
    h[30] = h[31] = h[32] = h[33] = 0xff;
    h[34] = h[36] = (unsigned char)(port >> 8);
    h[35] = h[37] = (unsigned char)(port & 0xff);
    h[38] = (unsigned char)(udp_len >> 8), h[39] = (unsigned char)(udp_len & 0xff);
}
}  // namespace

// ---------------------------------------------------------------------------------- reader
bool PacketFileReader::open(const std::string& filename)
{
    if (f_ && filename == fileName_) return true;  // vtkPacketFileReader.h:90-92
    close();
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) {
        lastError_ = filename + ": cannot open";
        return false;
    }
    unsigned char gh[24];
    if (std::fread(gh, 1, 24, f) != 24) {
        std::fclose(f);
        lastError_ = filename + ": truncated dump file";
        return false;
    }
    const uint32_t magic = rd32(gh, false);
    swap_ = nano_ = false;
    if (magic == 0xa1b2c3d4u) {
    } else if (magic == 0xa1b23c4du) {
        nano_ = true;
    } else if (magic == 0xd4c3b2a1u) {
        swap_ = true;
    } else if (magic == 0x4d3cb2a1u) {
        swap_ = nano_ = true;
    } else {
        std::fclose(f);
        lastError_ = filename + ": bad dump file format";
        return false;
    }
    f_ = f;
    fileName_ = filename;
    return true;
}

void PacketFileReader::close()
{
    if (f_) std::fclose(f_);
    f_ = nullptr;
    fileName_.clear();
}

void PacketFileReader::getFilePosition(int64_t* position) const
{
    if (position) *position = f_ ? (int64_t)std::ftell(f_) : -1;
}

void PacketFileReader::setFilePosition(const int64_t* position)
{
    if (f_ && position && *position >= 24) std::fseek(f_, (long)*position, SEEK_SET);
}

bool PacketFileReader::nextPacket(const unsigned char*& data, unsigned int& dataLength, int64_t& t_us)
{
    if (!f_) return false;
    for (;;) {
        unsigned char rh[16];
        const long at = std::ftell(f_);
        const size_t got = std::fread(rh, 1, 16, f_);
        if (got == 0) break;  // a clean end of file
        // a damaged capture must not look like a clean end: say where it broke
        auto damaged = [&](const char* what) {
            lastError_ = fileName_ + ": " + what + " at offset " + std::to_string(at);
        };
        if (got != 16) {
            damaged("truncated record header");
            break;
        }
        const uint32_t sec = rd32(rh, swap_), frac = rd32(rh + 4, swap_), incl = rd32(rh + 8, swap_);
        if (incl > (1u << 20)) {
            damaged("corrupt record (captured length beyond 1 MiB)");
            break;
        }
        rec_.resize(incl);
        if (incl && std::fread(rec_.data(), 1, incl, f_) != incl) {
            damaged("truncated record");
            break;
        }
        // the "udp" filter: Ethernet II, IPv4 WITHOUT options (IHL = 5: the fixed 42-byte strip of
        // vtkPacketFileReader.h:166-197 is only a UDP payload then), protocol 17; VLAN-tagged frames and
        // IPv4 options are skipped -- no HDL sensor or the reference's writer produces them
        if (incl < 42 || rec_[12] != 0x08 || rec_[13] != 0x00 || (rec_[14] & 0x0f) != 5 || rec_[23] != 17) continue;
        data = rec_.data() + 42;
        dataLength = incl - 42;
        t_us = (int64_t)sec * 1000000 + (nano_ ? frac / 1000 : frac);
        return true;
    }
    close();  // vtkPacketFileReader.h:176-180: the end of the file closes the reader
    return false;
}

// ---------------------------------------------------------------------------------- writer
bool PacketFileWriter::open(const std::string& filename)
{
    close();
    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) {
        lastError_ = "Failed to open packet file: " + filename;
        return false;
    }
    unsigned char gh[24];
    wr32(gh, 0xa1b2c3d4u);            // microsecond time stamps, this byte order
    gh[4] = 2, gh[5] = 0, gh[6] = 4, gh[7] = 0;  // version 2.4
    wr32(gh + 8, 0), wr32(gh + 12, 0);
    wr32(gh + 16, 65535);             // snap length (pcap_open_dead(DLT_EN10MB, 65535))
    wr32(gh + 20, 1);                 // Ethernet
    if (std::fwrite(gh, 1, 24, f) != 24) {
        std::fclose(f);
        lastError_ = "Failed to write packet file: " + filename;
        return false;
    }
    f_ = f;
    fileName_ = filename;
    return true;
}

void PacketFileWriter::close()
{
    if (f_) std::fclose(f_);
    f_ = nullptr;
    fileName_.clear();
}

bool PacketFileWriter::writePacket(const unsigned char* data, unsigned int dataLength, int64_t t_us)
{
    if (!f_ || !data) return false;
    unsigned char h[42];
    if (dataLength == 1206)
        prefix(h, 2368, 1214);
    else if (dataLength == 554 - 42)
        prefix(h, 8308, 520);
    else
        return false;  // vtkPacketFileWriter.cxx:141-144
    if (t_us == VELO_TIME_INVALID)
        t_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::system_clock::now().time_since_epoch())
                   .count();
    unsigned char rh[16];
    wr32(rh, (uint32_t)(t_us / 1000000));
    wr32(rh + 4, (uint32_t)(t_us % 1000000));
    wr32(rh + 8, dataLength + 42);
    wr32(rh + 12, dataLength + 42);
    return std::fwrite(rh, 1, 16, f_) == 16 && std::fwrite(h, 1, 42, f_) == 42 &&
           std::fwrite(data, 1, dataLength, f_) == dataLength;
}

}  // namespace veloslam
