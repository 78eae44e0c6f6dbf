// MapPatch.hpp -- a square map tile.  The reference's MapPatch (MapPatch.h:7-17)
// carries only vector features (posts / planes / line marks / complexes,
// MapObjects.h:11-46) and is never filled by any code; the north star registers
// frames against "the accumulated MapPatch cloud", so this MapPatch keeps the
// reference's fields (range, centerX, centerY) and gains a point payload.
#pragma once
#include <cstddef>
#include <vector>

namespace veloslam {

struct MapPatch {
    MapPatch(double x = 0, double y = 0, float r = 0) : range(r), centerX(x), centerY(y) {}
    float range;
    double centerX, centerY;
    std::vector<float> x, y, z;  // accumulated cloud of this tile (map frame, float32)
    size_t size() const { return x.size(); }
    void append(const float* px, const float* py, const float* pz, size_t n)
    {
        x.insert(x.end(), px, px + n);
        y.insert(y.end(), py, py + n);
        z.insert(z.end(), pz, pz + n);
    }
};

}  // namespace veloslam
