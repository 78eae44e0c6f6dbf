// HDLFrame.hpp -- one LiDAR revolution (HDLFrame.h:13-83), stored struct-of-arrays
// and beam-major so it uploads to HBM as-is: x[], y[], z[], intensity[] hold the
// beams back to back in vertical order (after the HDL-64 beam LUT,
// HDLParser.cxx:880-893) and beamStart[b]..beamStart[b+1] delimits beam b.
// getPointsAsOneCloud(start,end) (HDLFrame.cxx:127-144) is therefore a zero-copy
// view.  The advisory intrusive refcount (HDLFrame.cxx:211-219) is kept, atomic.
#pragma once
#include <atomic>
#include <cstdint>
#include <iosfwd>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "PoseTransform.hpp"

namespace veloslam {

struct PointMeta {  // type_defs.h:168-176
    unsigned short azimuth;
    float distance;
    unsigned char intensityFlag, distanceFlag, flags;
};

struct CloudView {  // what pcl::PointCloud<PointXYZI>::Ptr was to the reference's callers
    const float* x;
    const float* y;
    const float* z;
    const float* intensity;
    size_t size;
};

struct HDLFrame {
    HDLFrame();
    int64_t timestamp;  // microseconds
    std::vector<float> x, y, z, intensity;
    std::vector<uint16_t> packetIndex;  // per point: which packet of `packets` produced it
    std::vector<PointMeta> pointsMeta;
    std::vector<int32_t> beamStart;     // 65 entries for 64 beams
    std::vector<std::pair<int64_t, std::string>> packets;
    std::shared_ptr<PoseTransform> carpose;
    bool isInMemory, isOnHardDrive;
    std::atomic<unsigned char> count;
    // where the frame lives in its capture (HDLFrame.h:37-46): name of the file, offset of the
    // record that holds its first firing block, and how many blocks of that packet belong to the
    // previous frame.  firstPacket / numPackets (not in the reference) say the same in packets of
    // the capture as velo_pcap_read returns it: numPackets includes the packet that closes the frame.
    int64_t filenameTime, fileStartPos;
    int64_t firstPacket;
    int32_t numPackets;
    uint8_t skips;

    int numBeams() const { return beamStart.empty() ? 0 : (int)beamStart.size() - 1; }
    size_t numPoints() const { return x.size(); }
    // beams [startBeam, endBeam); like the reference, endBeam <= startBeam+1 yields
    // the single beam startBeam (HDLFrame.cxx:133-136)
    CloudView getPointsAsOneCloud(int startBeam = 0, int endBeam = 64) const;
    void setPoints(const float* px, const float* py, const float* pz, const float* pi,
                   const uint16_t* pkt, const int32_t* beam_start, int n_beams);
    void clear();  // HDLFrame.cxx:146-158
    // the .hdlmeta record (operator<< / operator>>, HDLFrame.cxx:160-190): timestamp, filenameTime,
    // fileStartPos, skips, isOnHardDrive, then the car pose record (type_defs.cxx:4-33).  ptime ->
    // int64 microseconds; fileStartPos keeps glibc's 16-byte fpos_t (offset + a zero shift state):
    // 132 bytes per frame.
    // the debug dumps (HDLFrame.cxx:36-125): `<dir>/<stamp>-points.txt` (x, y, z, intensity per line,
    // 9 significant digits, beams back to back), `-pointsMeta.txt` (azimuth, distance, three flags),
    // `-others.txt` (car pose, memory / disk state, file position, skips); and one beam (or, beamId
    // outside 0..63, all but the last -- the reference's range) as an ASCII .pcd with the fields
    // x y z intensity, the way pcl::io::savePCDFileASCII lays a PointXYZI cloud out.  <stamp> is
    // boost's to_iso_string of the time stamp (YYYYMMDDTHHMMSS[.ffffff]).  dumpToImage (OpenCV) is
    // not carried over.
    bool dumpToFiles(const std::string& dirname) const;
    bool dumpToPCD(const std::string& dirname, int beamId = -1) const;
    static std::string isoString(int64_t t_us);
    static constexpr size_t kMetaBytes = 132;
    bool writeMeta(std::ostream& os) const;
    bool readMeta(std::istream& is);
};

void intrusive_ptr_add_ref(HDLFrame* p);
void intrusive_ptr_release(HDLFrame* p);

}  // namespace veloslam
