// PacketFile.hpp -- the reference's capture reader and writer (vtkPacketFileReader.h:74-200,
// vtkPacketFileWriter.h:45-82 / .cxx:59-161) without libpcap: same method names, same container
// (24-byte global header, 16-byte record header, 42-byte Ethernet / IPv4 / UDP prefix in front of
// every payload -- PCAP_GLOBAL_HEADER_LEN / PCAP_PACKET_LEN of the reference).  Times are epoch
// microseconds as stored in the file (the reference's nextPacket adds its fixed + 8 h on the way to a
// ptime, type_defs.cxx:69-72; HDLManager::kClockShiftUs is that shift).  File positions are plain
// 64-bit offsets (the reference's fpos_t carries nothing else on Linux).
//
// velo_pcap_read / velo_pcap_write / velo_pcap_index (velo.h) are the bulk forms; these classes
// are for code that walks a capture record by record, or appends to one as packets arrive.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../velo.h"

namespace veloslam {

class PacketFileReader {
public:
    PacketFileReader() = default;
    ~PacketFileReader() { close(); }
    // true if `filename` is (already) open; every UDP datagram of the file is returned ("udp" filter)
    bool open(const std::string& filename);
    bool isOpen() const { return f_ != nullptr; }
    void close();
    const std::string& getLastError() const { return lastError_; }
    const std::string& getFileName() const { return fileName_; }
    void getFilePosition(int64_t* position) const;
    void setFilePosition(const int64_t* position);
    // the next UDP datagram: data -> its payload (the 42-byte prefix stripped; valid until the next
    // call), dataLength its size (1206 = lidar, 512 = position), t_us its record time.  false at the
    // end of the file (the reader closes itself, like the reference) or on a damaged record.
    bool nextPacket(const unsigned char*& data, unsigned int& dataLength, int64_t& t_us);

    PacketFileReader(const PacketFileReader&) = delete;
    void operator=(const PacketFileReader&) = delete;

private:
    FILE* f_ = nullptr;
    bool swap_ = false, nano_ = false;
    std::string fileName_, lastError_;
    std::vector<unsigned char> rec_;
};

class PacketFileWriter {
public:
    PacketFileWriter() = default;
    ~PacketFileWriter() { close(); }
    bool open(const std::string& filename);  // truncates; writes the global header
    bool isOpen() const { return f_ != nullptr; }
    void close();
    const std::string& GetLastError() const { return lastError_; }
    const std::string& GetFileName() const { return fileName_; }
    // a lidar packet (1206 bytes: prefix with ports 2368) or a position packet (512 bytes: ports
    // 8308) -- anything else is refused, as in the reference.  t_us = VELO_TIME_INVALID: now.
    bool writePacket(const unsigned char* data, unsigned int dataLength, int64_t t_us = VELO_TIME_INVALID);

    PacketFileWriter(const PacketFileWriter&) = delete;
    void operator=(const PacketFileWriter&) = delete;

private:
    FILE* f_ = nullptr;
    std::string fileName_, lastError_;
};

}  // namespace veloslam
