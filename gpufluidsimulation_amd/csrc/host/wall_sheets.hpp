// wall_sheets.hpp -- which cells the border taps of the compensation can touch, and who owns them.
//
// The reference zeroes the border nodes of the backward map in every DMC update (GPU_Advection.h:464-468, the
// protecting pre-copy is commented out at :335-337).  Stage 3 of gpu_compensate_* (cumulate_kernel with
// the backward map, GPU_kernel.cu:659-661, index window 2+d .. nb-3) interpolates the map at the node's nine sample
// points; on the first and last layers of the window some taps pair a live node with a zeroed one, and because ALL
// THREE map components of a border node are zero, such a tap's mapped position is
//         p' = s * q,      s = product over the axes of the live node's lerp weight (1/4, 1/2, 3/4),
// q being a convex combination of the live nodes' map values, i.e. within Dback + 1 cells of the tap.  On one GPU that
// merely samples the error field somewhere else; on a z-slab rank "somewhere else" is s * z, arbitrarily far below the
// slab.  This file computes, from nothing but the grid, the slab split and the displacement bound Dback (all identical
// on every rank), the boxes of the sampled field each rank's wall layers can touch, and cuts them by the rank that
// owns the planes -- so every rank knows what to send to whom and what to expect, with no negotiation.
//
// Exactness: boxes are conservative hulls (positions bounded by Dback, the bound the ghost-plane bookkeeping already
// relies on); flat-index wrap-around at a row's or plane's end (a position clamped to the top of the grid) is followed
// into the next row / plane, as the reference's flat indexing does.
#pragma once
#include <string>
#include <vector>
#include "bimocq_gpu.h"

namespace bqhost {

struct WallPlan {
    // wall indices (buffer index space; z: GLOBAL plane) whose taps interpolate towards a zeroed map node
    std::vector<int> xlist, ylist, zlist;
    int shadow_k0 = 0, shadow_k1 = 0;           // global buffer planes [k0, k1) the assembled copy must span (rank `rank`)
    std::vector<fl_box> local;                  // pieces this rank reads from its own local planes
    std::vector<std::vector<fl_box>> send;      // [peer]: pieces of MY owned planes that peer needs (in peer's order)
    std::vector<std::vector<fl_box>> recv;      // [peer]: pieces of peer's owned planes that I need
    // the same, flattened peer after peer (one gather / scatter launch per field), and everything the copy receives
    std::vector<fl_box> send_all, recv_all, placed_all;
    std::vector<size_t> send_vol, recv_vol;     // [peer]: floats
    bool empty() const { return xlist.empty() && ylist.empty() && zlist.empty(); }
    static size_t volume(const std::vector<fl_box> &v)
    {
        size_t n = 0;
        for (const fl_box &b : v) n += (size_t)(b.x1 - b.x0) * (size_t)(b.y1 - b.y0) * (size_t)(b.z1 - b.z0);
        return n;
    }
};

// ni, nj, nkg: GLOBAL cell dims; (dx, dy, dz): stagger of the sampled field; rank r owns the cell planes
// [r*nkg/nranks, (r+1)*nkg/nranks) and stores G ghost planes per side; Dback: displacement bound of the backward map in
// cells; need: ghost planes of the sampled field that are guaranteed correct locally (reachField(Dback)).
WallPlan make_wall_plan(int ni, int nj, int nkg, int dx, int dy, int dz, int rank, int nranks, int G, int Dback, int need);

} // namespace bqhost
