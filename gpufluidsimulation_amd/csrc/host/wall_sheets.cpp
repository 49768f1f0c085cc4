// wall_sheets.cpp -- see wall_sheets.hpp.
#include "wall_sheets.hpp"

#include <algorithm>
#include <cmath>

namespace bqhost {
namespace {

// one grid axis as cumulate_kernel sees it: n cells (z: global), stagger d, nb = n + d buffer entries, index window
// [L, H] = [2 + d, nb - 3] (GPU_kernel.cu:386-388)
struct Axis { int n, d, nb, L, H; };
Axis mk_axis(int n, int d) { return Axis{ n, d, n + d, 2 + d, n + d - 3 }; }

// map node `idx` along this axis keeps its value through the DMC update (DMC_backward_kernel writes 2 <= idx <= n - 3,
// GPU_kernel.cu:175; the rest of the cleared scratch is copied into the map, GPU_Advection.h:464-468)
inline bool live(const Axis &A, int idx) { return idx > 1 && idx < A.n - 2; }

// weight carried by the live node(s) of the map lerp of the tap at offset t (in cells) of buffer index w
double tap_factor(const Axis &A, int w, double t)
{
    const double m = (double)w - 0.5 * A.d + t;          // map coordinate in cells: multiples of 1/4, exact
    const int c = (int)std::floor(m);
    const double f = m - c;
    return (live(A, c) ? 1.0 - f : 0.0) + (live(A, c + 1) ? f : 0.0);
}

struct WallIdx { int w; std::vector<double> f; bool has_one; };

// window indices in [lo, hi] that have at least one tap with a factor != 1
std::vector<WallIdx> wall_indices(const Axis &A, int lo, int hi)
{
    std::vector<WallIdx> out;
    std::vector<int> cand;
    for (int w : { A.L, A.L + 1, A.L + 2, A.H - 2, A.H - 1, A.H })
        if (w >= A.L && w <= A.H && w >= lo && w <= hi && std::find(cand.begin(), cand.end(), w) == cand.end()) cand.push_back(w);
    std::sort(cand.begin(), cand.end());
    for (int w : cand) {
        WallIdx wi{ w, {}, false };
        for (double t : { -0.25, 0.25, 0.0 }) {
            const double fac = tap_factor(A, w, t);
            if (fac == 1.0) wi.has_one = true;
            else if (std::find(wi.f.begin(), wi.f.end(), fac) == wi.f.end()) wi.f.push_back(fac);
        }
        if (!wi.f.empty()) out.push_back(wi);
    }
    return out;
}

struct Opt { double tlo, thi, s; };                     // tap coordinates (cells) of a node set along one axis, lerp factor

// source index range [i0, i1] (inclusive, i1 may exceed nb - 1: flat-index wrap) of positions s * q, q within margin of
// the taps, clamped to [0, n] like cumulate_kernel's clampv3 (GPU_kernel.cu:395)
void index_range(const Axis &A, const Opt &o, double s, double margin, int &i0, int &i1)
{
    const double plo = std::min(std::max(s * (o.tlo - margin), 0.0), (double)A.n);
    const double phi = std::min(std::max(s * (o.thi + margin), 0.0), (double)A.n);
    i0 = (int)std::floor(plo + 0.5 * A.d);
    i1 = (int)std::floor(phi + 0.5 * A.d) + 1;
}

// inclusive box, x1 / y1 possibly past the row / plane end: follow the flat index into the next row / plane
void emit(std::vector<fl_box> &out, int x0, int x1, int y0, int y1, int z0, int z1, int nbx, int nby, int nbz)
{
    if (x0 > x1 || y0 > y1 || z0 > z1 || z0 >= nbz) return;
    if (x1 >= nbx) {
        emit(out, x0, nbx - 1, y0, y1, z0, z1, nbx, nby, nbz);
        emit(out, 0, x1 - nbx, y0 + 1, y1 + 1, z0, z1, nbx, nby, nbz);
        return;
    }
    if (y1 >= nby) {
        emit(out, x0, x1, y0, nby - 1, z0, z1, nbx, nby, nbz);
        emit(out, x0, x1, 0, y1 - nby, z0 + 1, z1 + 1, nbx, nby, nbz);
        return;
    }
    out.push_back(fl_box{ x0, x1 + 1, y0, y1 + 1, std::max(z0, 0), std::min(z1 + 1, nbz) });    // beyond the last plane reads 0
}

bool contains(const fl_box &a, const fl_box &b)          // a contains b
{
    return a.x0 <= b.x0 && b.x1 <= a.x1 && a.y0 <= b.y0 && b.y1 <= a.y1 && a.z0 <= b.z0 && b.z1 <= a.z1;
}

struct Piece { int source; fl_box box; };

} // namespace

WallPlan make_wall_plan(int ni, int nj, int nkg, int dx, int dy, int dz, int rank, int nranks, int G, int Dback, int need)
{
    WallPlan plan;
    plan.send.assign((size_t)nranks, {});
    plan.recv.assign((size_t)nranks, {});
    const Axis X = mk_axis(ni, dx), Y = mk_axis(nj, dy), Z = mk_axis(nkg, dz);
    const double margin = (double)Dback + 1.0;           // q: convex combination of live nodes within one cell of the tap
    auto own0 = [&](int r) { return (int)((long long)r * nkg / nranks); };
    auto own1 = [&](int r) { return (int)((long long)(r + 1) * nkg / nranks); };
    auto owner = [&](int z) {                            // of buffer plane z (w-type buffers: the top plane goes with the last rank)
        if (z >= nkg) return nranks - 1;
        int r = (int)(((long long)z * nranks) / nkg);
        while (r > 0 && z < own0(r)) r--;
        while (r < nranks - 1 && z >= own1(r)) r++;
        return r;
    };
    const std::vector<WallIdx> wx = wall_indices(X, X.L, X.H), wy = wall_indices(Y, Y.L, Y.H);

    std::vector<Piece> mine;                             // pieces of rank `rank`'s need, in order
    for (int r = 0; r < nranks; r++) {
        // destination planes of rank r: what the stage may produce locally (owned + G ghost planes), inside the window
        const int zlo = std::max(Z.L, own0(r) - G), zhi = std::min(Z.H, own1(r) + G - 1 + dz);
        if (zlo > zhi) continue;
        const std::vector<WallIdx> wz = wall_indices(Z, zlo, zhi);
        if (r == rank) {
            for (const WallIdx &w : wx) plan.xlist.push_back(w.w);
            for (const WallIdx &w : wy) plan.ylist.push_back(w.w);
            for (const WallIdx &w : wz) plan.zlist.push_back(w.w);
        }
        // per axis: the node sets a wall layer can combine with (the whole window at factor 1, or another wall layer at
        // one of its factors), and the wall layers themselves
        const Axis *ax[3] = { &X, &Y, &Z };
        const std::vector<WallIdx> *walls[3] = { &wx, &wy, &wz };
        const double wlo[3] = { (double)X.L, (double)Y.L, (double)zlo }, whi[3] = { (double)X.H, (double)Y.H, (double)zhi };
        std::vector<Opt> other[3], self_opts;
        for (int a = 0; a < 3; a++) {
            const double off = 0.5 * ax[a]->d;
            other[a].push_back(Opt{ wlo[a] - off - 0.25, whi[a] - off + 0.25, 1.0 });
            for (const WallIdx &w : *walls[a])
                for (double f : w.f) other[a].push_back(Opt{ w.w - off - 0.25, w.w - off + 0.25, f });
        }
        std::vector<fl_box> boxes;
        for (int a = 0; a < 3; a++)
            for (const WallIdx &w : *walls[a]) {
                const double off = 0.5 * ax[a]->d;
                self_opts.clear();
                for (double f : w.f) self_opts.push_back(Opt{ w.w - off - 0.25, w.w - off + 0.25, f });
                if (w.has_one) self_opts.push_back(Opt{ w.w - off - 0.25, w.w - off + 0.25, 1.0 });
                const int b = (a + 1) % 3, c = (a + 2) % 3;
                for (const Opt &oa : self_opts)
                    for (const Opt &ob : other[b])
                        for (const Opt &oc : other[c]) {
                            const double s = oa.s * ob.s * oc.s;
                            int lo[3], hi[3];
                            index_range(*ax[a], oa, s, margin, lo[a], hi[a]);
                            index_range(*ax[b], ob, s, margin, lo[b], hi[b]);
                            index_range(*ax[c], oc, s, margin, lo[c], hi[c]);
                            emit(boxes, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], X.nb, Y.nb, Z.nb);
                        }
            }
        // drop boxes another one covers (first occurrence wins among equals)
        std::vector<fl_box> kept;
        for (size_t i = 0; i < boxes.size(); i++) {
            bool covered = false;
            for (size_t j = 0; j < boxes.size() && !covered; j++) {
                if (i == j) continue;
                if (contains(boxes[j], boxes[i])) {
                    const bool same = contains(boxes[i], boxes[j]);
                    covered = !same || j < i;
                }
            }
            if (!covered && boxes[i].x1 > boxes[i].x0 && boxes[i].y1 > boxes[i].y0 && boxes[i].z1 > boxes[i].z0) kept.push_back(boxes[i]);
        }
        // cut along z by where the planes come from: rank r's own correct local planes, else the owner
        const int llo = std::max(0, own0(r) - need), lhi = std::min(Z.nb, own1(r) + need + dz);
        for (const fl_box &bx : kept) {
            int z = bx.z0;
            while (z < bx.z1) {
                const int src = (z >= llo && z < lhi) ? r : owner(z);
                int e = z + 1;
                while (e < bx.z1 && ((e >= llo && e < lhi) ? r : owner(e)) == src) e++;
                fl_box p = bx; p.z0 = z; p.z1 = e;
                if (r == rank) mine.push_back(Piece{ src, p });
                else if (src == rank) plan.send[(size_t)r].push_back(p);
                z = e;
            }
        }
    }
    bool first = true;
    for (const Piece &p : mine) {
        if (p.source == rank) plan.local.push_back(p.box); else plan.recv[(size_t)p.source].push_back(p.box);
        if (first || p.box.z0 < plan.shadow_k0) plan.shadow_k0 = p.box.z0;
        if (first || p.box.z1 > plan.shadow_k1) plan.shadow_k1 = p.box.z1;
        first = false;
    }
    plan.send_vol.assign((size_t)nranks, 0);
    plan.recv_vol.assign((size_t)nranks, 0);
    plan.placed_all = plan.local;
    for (int q = 0; q < nranks; q++) {
        plan.send_vol[(size_t)q] = WallPlan::volume(plan.send[(size_t)q]);
        plan.recv_vol[(size_t)q] = WallPlan::volume(plan.recv[(size_t)q]);
        plan.send_all.insert(plan.send_all.end(), plan.send[(size_t)q].begin(), plan.send[(size_t)q].end());
        plan.recv_all.insert(plan.recv_all.end(), plan.recv[(size_t)q].begin(), plan.recv[(size_t)q].end());
        plan.placed_all.insert(plan.placed_all.end(), plan.recv[(size_t)q].begin(), plan.recv[(size_t)q].end());
    }
    return plan;
}

} // namespace bqhost
