// gpu_mapper.cpp -- see gpu_mapper.hpp.  Reference: src/bimocq3D/GPU_Advection.h:110-627.
#include "gpu_mapper.hpp"
#include <algorithm>
#include <cstdio>

#include <cstdlib>

namespace bqhost {

void (*g_trace_hook)(const char *stage) = nullptr;

gpuMapper::gpuMapper(int device, int nx, int ny, int nz, float h, const SlabCtx &sl)
{
    slab = sl;
    g.ni = nx; g.nj = ny; g.nk = slab.on ? slab.nk_local() : nz; g.h = h;
    if (fl_init(device) != FL_OK) return;                 // cudaInit(), GPU_Advection.h:214-226
    // The operator ABI's size limits (include/bimocq_gpu.h, "Limits"), checked before anything is allocated: the gather
    // kernels address a field through a buffer descriptor with 32-bit byte offsets, so ONE field -- (ni+1)(nj+1)(nk+1)
    // floats at most -- must stay below 2 GiB, and planes are launched along grid.z (nk + 1 <= 65 535).  BASELINE config 5
    // (1024 x 1024 x 512: 2.0 GiB per field) therefore needs at least two z-slab ranks.
    {
        const double per_plane = 4.0 * (double)(g.ni + 1) * (double)(g.nj + 1);
        if (per_plane * (double)(g.nk + 1) >= 2147483648.0 || g.nk + 1 > 65535) {
            int ranks = 0;
            const int G = slab.on && slab.G > 0 ? slab.G : 8;
            for (int r = 2; r <= 4096 && !ranks; r++)
                if (nz % r == 0 && per_plane * (double)(nz / r + 2 * G + 1) < 2147483648.0 && nz / r + 2 * G + 1 <= 65535) ranks = r;
            char msg[240];
            snprintf(msg, sizeof msg, "a field of %d x %d x %d planes is %.2f GiB: the operators take fields below 2 GiB (32-bit buffer "
                     "descriptors) and at most 65534 planes; split the grid into at least %d z-slab ranks",
                     g.ni, g.nj, g.nk, per_plane * (double)(g.nk + 1) / 1073741824.0, ranks);
            fl_report_error(FL_ERR_UNSUPPORTED, msg);
            return;
        }
    }
    if (slab.on) fl_set_slab(slab.koff(), slab.nkg, slab.own0, slab.own1, slab.nk_local());
    else fl_set_slab(0, 0, 0, 0, 0);
    ok_ = allocField(u_src, FIELD_U) && allocField(v_src, FIELD_V) && allocField(w_src, FIELD_W)
       && allocField(x_out, FIELD_S) && allocField(y_out, FIELD_S) && allocField(z_out, FIELD_S)
       && allocField(x_out2, FIELD_S) && allocField(y_out2, FIELD_S) && allocField(z_out2, FIELD_S);
}

bool gpuMapper::allocField(DeviceField &f, FieldKind kind) const
{
    size_t count = kind == FIELD_U ? g.nu() : kind == FIELD_V ? g.nv() : kind == FIELD_W ? g.nw() : g.n();
    if (!f.alloc(count)) return false;
    f.plane = kind == FIELD_U ? (size_t)(g.ni + 1) * g.nj : kind == FIELD_V ? (size_t)g.ni * (g.nj + 1) : (size_t)g.ni * g.nj;
    f.extra = kind == FIELD_W ? 1 : 0;
    f.valid = slab.on ? slab.G : DeviceField::kAlwaysValid;      // zero-filled: trivially consistent
    return true;
}

void gpuMapper::require(std::initializer_list<DeviceField *> fields, int depth)
{
    if (!slab.on || slab.nranks <= 1) return;
    if (depth > slab.G) {
        // the caller asked for more ghost planes than exist: the time step moved data further than the
        // slab overlap (CFL too large for G); refuse rather than compute from missing planes
        fl_report_error(FL_ERR_BAD_ARGUMENT, "z-slab ghost zone too shallow for this time step (CFL travel + stencil reach > G)");
        return;
    }
    // BQ_HALO_ALWAYS=1 (debug): refresh on every request, whatever the bookkeeping says
    static const bool always = getenv("BQ_HALO_ALWAYS") && atoi(getenv("BQ_HALO_ALWAYS")) != 0;
    float *ptrs[16]; size_t planes[16]; int extras[16]; int n = 0;
    for (DeviceField *f : fields)
        if ((always || f->valid < depth) && n < 16) { ptrs[n] = f->get(); planes[n] = f->plane; extras[n] = f->extra; n++; }
    if (!n) return;
    // BQ_TRACE_REQUIRE=1 (debug): what triggered the exchange -- requested depth and the shallowest valid depth found
    static const bool trace = getenv("BQ_TRACE_REQUIRE") && atoi(getenv("BQ_TRACE_REQUIRE")) != 0;
    if (trace && slab.rank == 0) {
        int vmin = 1 << 20;
        for (DeviceField *f : fields) if (f->valid < vmin) vmin = f->valid;
        fprintf(stderr, "[require] %d of %d fields, depth %d requested, valid %d\n", n, (int)fields.size(), depth, vmin);
    }
    // All G planes move, whatever depth was asked for: moving only `depth` planes was measured on two ranks
    // (tests/slab_worker.py, 32x32x64, G = 8): 26 % fewer planes outside the Jacobi loop but 4 more exchanges per step
    // (the deeper validity is what spares later operators their own exchange) -- no gain.
    // (BQ_OPT_SHALLOW_BLOCKING_EXCHANGE: only `depth` planes -- what pays depends on the links, see include/bimocq_solver.h)
    const int moved = shallow_blocking ? depth : slab.G;
    fl_halo_exchange(n, ptrs, planes, extras, g.nk, slab.G, moved, 1);
    for (DeviceField *f : fields)
        if (always || f->valid < depth) f->valid = moved;
}

const WallPlan &gpuMapper::wallPlan(FieldKind kind, int Dback, int need)
{
    const auto key = std::make_tuple((int)kind, Dback, need);
    auto it = wall_plans_.find(key);
    if (it == wall_plans_.end())
        it = wall_plans_.emplace(key, make_wall_plan(g.ni, g.nj, slab.nkg, kind == FIELD_U, kind == FIELD_V, kind == FIELD_W,
                                                     slab.rank, slab.nranks, slab.G, Dback, need)).first;
    return it->second;
}

void gpuMapper::wallFixup(std::initializer_list<WallItem> items, DeviceField &bx, DeviceField &by, DeviceField &bz,
                          int Dback, int need, float coeff, int out_valid)
{
    wallFixupBegin(items, Dback, need);
    wallFixupEnd(bx, by, bz, coeff, out_valid);
}

// First half: the pieces other ranks need are gathered from MY OWNED planes (always correct) and the messages start on
// the halo stream -- before the stage-3 operator they will correct is launched, so that they travel while it runs.
void gpuMapper::wallFixupBegin(std::initializer_list<WallItem> items, int Dback, int need)
{
    wall_n_ = 0;
    if (!slab.on || slab.nranks <= 1 || items.size() == 0 || items.size() > 3) return;
    const int nr = slab.nranks;
    int n = 0;
    for (const WallItem &it : items) {
        const WallPlan &p = wallPlan(it.kind, Dback, need);
        if (p.empty() || p.shadow_k1 <= p.shadow_k0) continue;
        wall_items_[n] = it; wall_plans_now_[n] = &p; n++;
    }
    if (!n) return;
    // the assembled copies: the global planes [k0, k1) of the sampled field, NaN wherever nothing has been placed.
    // Send / receive buffers: field after field, inside a field peer after peer (both sides derive the same order)
    size_t send_total = 0, recv_total = 0;
    for (int a = 0; a < n; a++) {
        const WallPlan &p = *wall_plans_now_[a];
        int nbi, nbj, nkf;
        wallDims(wall_items_[a].kind, nbi, nbj, nkf);
        const size_t plane = (size_t)nbi * (size_t)nbj;
        Shadow &sh = wall_shadow_[a];
        const size_t count = plane * (size_t)(p.shadow_k1 - p.shadow_k0);
        if (sh.buf.count() < count || sh.plane != plane || sh.k0 != p.shadow_k0 || sh.k1 != p.shadow_k1) {
            if (sh.buf.count() < count && !sh.buf.alloc(count)) return;
            fl_memset(sh.buf.get(), 0xFF, sh.buf.bytes());           // all-ones bytes: a NaN in every float
            sh.k0 = p.shadow_k0; sh.k1 = p.shadow_k1; sh.plane = plane;
        }
        wall_send_base_[a] = send_total; wall_recv_base_[a] = recv_total;
        send_total += WallPlan::volume(p.send_all);
        recv_total += WallPlan::volume(p.recv_all);
    }
    if ((wall_send_.count() < send_total && !wall_send_.alloc(send_total + send_total / 8 + 64)) ||
        (wall_recv_.count() < recv_total && !wall_recv_.alloc(recv_total + recv_total / 8 + 64))) return;
    for (int a = 0; a < n; a++) {
        const WallPlan &p = *wall_plans_now_[a];
        int nbi, nbj, nkf;
        wallDims(wall_items_[a].kind, nbi, nbj, nkf);
        if (!p.send_all.empty())
            fl_box_pack(wall_items_[a].src->get(), nbi, nbj, nkf, slab.koff(), p.send_all.data(), (int)p.send_all.size(),
                        wall_send_.get() + wall_send_base_[a]);
    }
    // one group of point-to-point messages: per field and peer that owes me planes or that I owe planes
    std::vector<int> peers; std::vector<float *> sp, rp; std::vector<size_t> sc, rc;
    for (int a = 0; a < n; a++) {
        const WallPlan &p = *wall_plans_now_[a];
        size_t so = wall_send_base_[a], ro = wall_recv_base_[a];
        for (int q = 0; q < nr; q++) {
            const size_t ns = p.send_vol[(size_t)q], nrv = p.recv_vol[(size_t)q];
            if (q != slab.rank && (ns || nrv)) {
                peers.push_back(q); sp.push_back(wall_send_.get() + so); sc.push_back(ns);
                rp.push_back(wall_recv_.get() + ro); rc.push_back(nrv);
            }
            so += ns; ro += nrv;
        }
    }
    if (!peers.empty()) fl_p2p_exchange_begin((int)peers.size(), peers.data(), sp.data(), sc.data(), rp.data(), rc.data());
    wall_bytes_moved += (long long)recv_total * 4;
    wall_n_ = n;
}

// Second half, after the stage-3 operator: my own pieces (they may lie on ghost planes the operator's exchange has just
// refreshed) and what arrived go into the copies, the wall layers are re-evaluated on the planes the stage produced, and
// the copies are blanked again (a cell the plan did not foresee must read NaN next time as well, never stale data).
void gpuMapper::wallFixupEnd(DeviceField &bx, DeviceField &by, DeviceField &bz, float coeff, int out_valid)
{
    const int n = wall_n_;
    wall_n_ = 0;
    if (!n) return;
    for (int a = 0; a < n; a++) {
        const WallPlan &p = *wall_plans_now_[a];
        int nbi, nbj, nkf;
        wallDims(wall_items_[a].kind, nbi, nbj, nkf);
        Shadow &sh = wall_shadow_[a];
        if (!p.local.empty())
            fl_box_copy(wall_items_[a].src->get(), nbi, nbj, nkf, slab.koff(), sh.buf.get(), sh.k1 - sh.k0, sh.k0, p.local.data(), (int)p.local.size());
    }
    fl_halo_wait();                                      // the messages of wallFixupBegin
    const bool can_window = overlap_exchanges && fuse_housekeeping && fl_get_option(FL_OPT_FUSED_HOUSEKEEPING) >= 0;
    const int ov = out_valid < 0 ? 0 : (out_valid > slab.G ? slab.G : out_valid);
    const int w0 = can_window ? slab.G - ov : 0, w1 = can_window ? g.nk - slab.G + (ov > 1 ? ov : 1) : g.nk;
    for (int a = 0; a < n; a++) {
        const WallPlan &p = *wall_plans_now_[a];
        int nbi, nbj, nkf;
        wallDims(wall_items_[a].kind, nbi, nbj, nkf);
        Shadow &sh = wall_shadow_[a];
        if (!p.recv_all.empty())
            fl_box_unpack(sh.buf.get(), nbi, nbj, sh.k1 - sh.k0, sh.k0, p.recv_all.data(), (int)p.recv_all.size(), wall_recv_.get() + wall_recv_base_[a]);
    }
    const bool windowed = can_window && (w0 > 0 || w1 < g.nk) && fl_set_plane_window(w0, w1) == 1;
    for (int a = 0; a < n; a++) {
        const WallPlan &p = *wall_plans_now_[a];
        const FieldKind kind = wall_items_[a].kind;
        const int axis = kind == FIELD_U ? 0 : kind == FIELD_V ? 1 : kind == FIELD_W ? 2 : -1;
        Shadow &sh = wall_shadow_[a];
        gpu_accumulate_wall_fixup(sh.buf.get(), sh.k0, sh.k1 - sh.k0, wall_items_[a].before->get(), wall_items_[a].dst->get(),
                                  bx.get(), by.get(), bz.get(), g.h, g.ni, g.nj, g.nk, axis, coeff,
                                  p.xlist.data(), (int)p.xlist.size(), p.ylist.data(), (int)p.ylist.size(),
                                  p.zlist.data(), (int)p.zlist.size());
    }
    if (windowed) fl_set_plane_window(-1, -1);
    for (int a = 0; a < n; a++) {
        const WallPlan &p = *wall_plans_now_[a];
        int nbi, nbj, nkf;
        wallDims(wall_items_[a].kind, nbi, nbj, nkf);
        Shadow &sh = wall_shadow_[a];
        if (!p.placed_all.empty())
            fl_box_unpack(sh.buf.get(), nbi, nbj, sh.k1 - sh.k0, sh.k0, p.placed_all.data(), (int)p.placed_all.size(), nullptr);
    }
}

// Whole-grid copies of slab fields: my owned planes by a device copy, every peer's by one point-to-point message per field, all
// in one group (xGMI is a full mesh).  Equal slabs (nkg / nranks planes each; a w-type field's top face goes with the last rank),
// as everywhere in the slab path.
bool gpuMapper::assembleGlobal(std::initializer_list<GlobalPair> fields)
{
    if (!slab.on || slab.nranks <= 1) return false;
    const int R = slab.nranks, nkg = slab.nkg, G = slab.G;
    const auto own0_of = [&](int r) { return r * (nkg / R); };
    const auto planes_of = [&](int r, int extra) { return nkg / R + ((extra && r == R - 1) ? 1 : 0); };
    std::vector<int> peers; std::vector<float *> send, recv; std::vector<size_t> send_count, recv_count;
    for (const GlobalPair &f : fields) {
        const size_t plane = f.local->plane, total = plane * (size_t)(nkg + f.local->extra);
        if (!plane) return false;
        if (f.global->count() < total && !f.global->alloc(total)) return false;
        fl_memcpy_d2d(f.global->get() + plane * (size_t)slab.own0, f.local->get() + plane * (size_t)G,
                      plane * (size_t)planes_of(slab.rank, f.local->extra) * sizeof(float));
        bool known = false;
        for (auto &t : global_twins_) if (t.first == f.local) { t.second = f.global; known = true; }
        if (!known) global_twins_.emplace_back(f.local, f.global);
    }
    for (int r = 0; r < R; r++) {
        if (r == slab.rank) continue;
        for (const GlobalPair &f : fields) {
            const size_t plane = f.local->plane;
            peers.push_back(r);
            send.push_back(f.local->get() + plane * (size_t)G);
            send_count.push_back(plane * (size_t)planes_of(slab.rank, f.local->extra));
            recv.push_back(f.global->get() + plane * (size_t)own0_of(r));
            recv_count.push_back(plane * (size_t)planes_of(r, f.local->extra));
            global_prev_bytes += (long long)recv_count.back() * 4;
        }
    }
    fl_p2p_exchange((int)peers.size(), peers.data(), send.data(), send_count.data(), recv.data(), recv_count.data());
    return fl_last_error() == FL_OK;
}

void gpuMapper::startEventRecord()
{
    if (!ev_start_) ev_start_ = fl_event_create();
    if (!ev_stop_) ev_stop_ = fl_event_create();
    fl_event_record(ev_start_);
}

float gpuMapper::endEventRecord()
{
    fl_event_record(ev_stop_);
    return fl_event_elapsed_ms(ev_start_, ev_stop_);
}

// GPU_Advection.h:472-482: the kernels write an interior window only, so zero first
void gpuMapper::advectVelocity(float *u, float *v, float *w, float *ui, float *vi, float *wi,
                               float *bx, float *by, float *bz, bool is_point) const
{
    FusedScope fused(fuse_housekeeping, 1);             // the kernels write the zeros outside their window themselves
    if (!fused.on) {
        fl_memset(u, 0, g.nu() * sizeof(float));
        fl_memset(v, 0, g.nv() * sizeof(float));
        fl_memset(w, 0, g.nw() * sizeof(float));
    }
    gpu_advect_velocity(u, v, w, ui, vi, wi, bx, by, bz, g.h, g.ni, g.nj, g.nk, is_point);
}

// GPU_Advection.h:493-503
void gpuMapper::compensateVelocity(float *u, float *v, float *w, float *du, float *dv, float *dw,
                                   float *fx, float *fy, float *fz, float *bx, float *by, float *bz,
                                   bool is_point) const
{
    fl_memset(u_src, 0, u_src.bytes()); fl_memset(v_src, 0, v_src.bytes()); fl_memset(w_src, 0, w_src.bytes());
    gpu_compensate_velocity(u, v, w, du, dv, dw, u_src, v_src, w_src, fx, fy, fz, bx, by, bz,
                            g.h, g.ni, g.nj, g.nk, is_point);
}

// GPU_Advection.h:505-511 (sized ni*nj*nk, not the reference's (ni+1)*nj*nk overrun)
void gpuMapper::advectField(float *f, float *fi, float *bx, float *by, float *bz, bool is_point) const
{
    FusedScope fused(fuse_housekeeping, 1);
    if (!fused.on) fl_memset(f, 0, g.n() * sizeof(float));
    gpu_advect_field(f, fi, bx, by, bz, g.h, g.ni, g.nj, g.nk, is_point);
}

// GPU_Advection.h:521-528: u_src doubles as the scalar error scratch
void gpuMapper::compensateField(float *f, float *df, float *fx, float *fy, float *fz,
                                float *bx, float *by, float *bz, bool is_point) const
{
    fl_memset(u_src, 0, g.n() * sizeof(float));
    gpu_compensate_field(f, df, u_src, fx, fy, fz, bx, by, bz, g.h, g.ni, g.nj, g.nk, is_point);
}

// GPU_Advection.h:602-608
void gpuMapper::projectionJacobi(float *u, float *v, float *w, float *div, float *p, float *p_temp, float *debug,
                                 int iter, float halfrdx, float alpha, float beta) const
{
    fl_memset(div, 0, g.n() * sizeof(float));
    fl_memset(p, 0, g.n() * sizeof(float));
    fl_memset(p_temp, 0, g.n() * sizeof(float));
    gpu_projection_jacobi(u, v, w, div, p, p_temp, debug, g.ni, g.nj, g.nk, iter, halfrdx, alpha, beta);
}

} // namespace bqhost
