// gpu_mapper.hpp -- host-side owner of device buffers and thin operator wrappers.
//
// Mirrors the role of the reference's `gpuMapper` (src/bimocq3D/GPU_Advection.h:110-627): same
// method names, argument order and pre-zeroing rules, so the solver code above it reads like the
// reference's call sites.  Differences, all deliberate:
//   * it speaks only the C-ABI of include/bimocq_gpu.h (fl_* instead of cuda*), so the same host
//     code links against the HIP library in the product and against a CPU stand-in in tests;
//   * buffers are released in the destructor (the reference never frees anything);
//   * the DMC sub-step ping-pongs between buffers instead of copying out -> in three times.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <initializer_list>
#include <map>
#include <tuple>
#include <utility>
#include <vector>
#include "bimocq_gpu.h"
#include "wall_sheets.hpp"

namespace bqhost {

// untyped zero-filled device allocation (the fp64 arrays of the multigrid projection), RAII
class DeviceBytes {
public:
    DeviceBytes() = default;
    DeviceBytes(const DeviceBytes &) = delete;
    DeviceBytes &operator=(const DeviceBytes &) = delete;
    DeviceBytes(DeviceBytes &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceBytes &operator=(DeviceBytes &&o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DeviceBytes() { release(); }
    bool alloc(size_t bytes) { release(); p_ = fl_malloc(bytes ? bytes : 8); n_ = p_ ? bytes : 0; return p_ != nullptr; }
    void release() { if (p_) fl_free(p_); p_ = nullptr; n_ = 0; }
    double *f64() const { return static_cast<double *>(p_); }
    size_t bytes() const { return n_; }
private:
    void *p_ = nullptr;
    size_t n_ = 0;
};

// allocGPUBuffer (GPU_Advection.h:322-326): zero-filled device allocation, RAII.
class DeviceField {
public:
    DeviceField() = default;
    explicit DeviceField(size_t count) { alloc(count); }
    DeviceField(const DeviceField &) = delete;
    DeviceField &operator=(const DeviceField &) = delete;
    DeviceField(DeviceField &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceField &operator=(DeviceField &&o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DeviceField() { release(); }

    bool alloc(size_t count)
    {
        release();
        p_ = static_cast<float *>(fl_malloc(count * sizeof(float)));
        n_ = p_ ? count : 0;
        return p_ != nullptr;
    }
    void release() { if (p_) fl_free(p_); p_ = nullptr; n_ = 0; }
    // an all-zero field is consistent across slab ranks; a copy inherits the source's ghost validity
    void zero() { fl_memset(p_, 0, bytes()); valid = kAlwaysValid; }
    void copy_from(const DeviceField &src)
    {
        fl_memcpy_d2d(p_, src.p_, bytes() < src.bytes() ? bytes() : src.bytes());
        valid = src.valid;
    }
    void download(float *host) const { fl_memcpy_d2h(host, p_, bytes()); }
    void upload(const float *host) const { fl_memcpy_h2d(p_, host, bytes()); }
    float *get() const { return p_; }
    operator float *() const { return p_; }
    size_t count() const { return n_; }
    size_t bytes() const { return n_ * sizeof(float); }
    void swap(DeviceField &o)
    {
        float *p = p_; size_t n = n_; p_ = o.p_; n_ = o.n_; o.p_ = p; o.n_ = n;
        int v = valid; valid = o.valid; o.valid = v;
    }

    // z-slab bookkeeping (multi-GPU): how many ghost planes per side currently hold correct values,
    // elements per plane, and 1 for a w-type buffer (nk+1 planes).  Ignored on a single GPU.
    int    valid = kAlwaysValid;
    size_t plane = 0;
    int    extra = 0;
    static constexpr int kAlwaysValid = 1 << 20;

private:
    float *p_ = nullptr;
    size_t n_ = 0;
};

// FL_OPT_FUSED_HOUSEKEEPING bits for the duration of a scope.  `on` is false when the caller does not want it or
// the operator library underneath does not implement the option (fl_get_option < 0: the CPU stand-in of the tests);
// the caller then issues the clears and copies itself.
struct FusedScope {
    int prev = -1;
    bool on = false;
    FusedScope(bool wanted, int bits)
    {
        if (!wanted) return;
        prev = fl_get_option(FL_OPT_FUSED_HOUSEKEEPING);
        on = prev >= 0;
        if (on) fl_set_option(FL_OPT_FUSED_HOUSEKEEPING, bits);
    }
    ~FusedScope() { if (on) fl_set_option(FL_OPT_FUSED_HOUSEKEEPING, prev); }
    FusedScope(const FusedScope &) = delete;
    FusedScope &operator=(const FusedScope &) = delete;
};

// z-slab decomposition of the global grid.  Rank `rank` of `nranks` owns the global cell
// planes [own0, own1) and stores [own0 - G, own1 + G); off = single GPU.
struct SlabCtx {
    bool on = false;
    int rank = 0, nranks = 1;
    int nkg = 0;            // global cell planes
    int own0 = 0, own1 = 0;
    int G = 0;              // ghost planes per side
    int koff() const { return own0 - G; }
    int nk_local() const { return own1 - own0 + 2 * G; }
};

struct GridDims {
    int ni = 0, nj = 0, nk = 0;     // nk: LOCAL cell planes (owned + ghosts on a slab rank)
    float h = 0.f;
    size_t n() const { return (size_t)ni * nj * nk; }
    size_t nu() const { return (size_t)(ni + 1) * nj * nk; }
    size_t nv() const { return (size_t)ni * (nj + 1) * nk; }
    size_t nw() const { return (size_t)ni * nj * (nk + 1); }
};

enum FieldKind { FIELD_U, FIELD_V, FIELD_W, FIELD_S };

// debug hook (BQ_TRACE): called with a stage name from inside the mapper sequences
extern void (*g_trace_hook)(const char *stage);
inline void trace_point(const char *stage) { if (g_trace_hook) g_trace_hook(stage); }

class gpuMapper {
public:
    // nz: GLOBAL cell planes; with a slab context the buffers hold slab.nk_local() planes
    gpuMapper(int device, int nx, int ny, int nz, float h, const SlabCtx &slab = SlabCtx());
    bool ok() const { return ok_; }

    GridDims g;
    SlabCtx slab;
    // allocate a field of the given staggering, with its slab metadata
    bool allocField(DeviceField &f, FieldKind kind) const;
    // make sure the listed fields have at least `depth` valid ghost planes: those that do not are
    // refreshed from the z-neighbours (all G planes, one RCCL group).  No-op on a single GPU.
    void require(std::initializer_list<DeviceField *> fields, int depth);
    // The same requirement for one or more (fields, depth) pairs followed by the operator `op` that consumes them, with
    // the exchange OVERLAPPED: the ghost planes travel on the halo stream while `op` runs on the planes that cannot
    // reach them (fl_set_plane_window: [G + reach, nk - G - reach)); after fl_halo_wait it runs on the two ends.
    // `op` is called up to three times and must consist of window-honouring operator calls only (no clears, no
    // copies: that is the case with the fused housekeeping) and must not write a field that is being exchanged.
    // Falls back to require() + op() when the operator library has no plane windows (CPU stand-in), the housekeeping
    // is not fused, or the slab is too thin.
    // out_valid (optional): how many ghost planes of the operator's outputs will be correct afterwards (what the caller
    // passes to produced()).  Nothing may read an output beyond that, so the operator is not run there at all: with
    // G = 8 and a typical out_valid of 4 it covers 264 instead of 272 planes.
    struct GhostNeed { std::initializer_list<DeviceField *> fields; int depth; };
    // writes: the fields `op` stores into.  None of them may be among the fields in flight (the halo stream reads their
    // owned planes and writes their ghost planes while `op` runs): checked, not assumed -- a violation latches an error
    // and falls back to exchange-then-operator.
    template <class Op>
    void withGhosts(std::initializer_list<GhostNeed> needs, Op &&op, int out_valid = DeviceField::kAlwaysValid,
                    std::initializer_list<const DeviceField *> writes = {})
    {
        if (!slab.on || slab.nranks <= 1) { op(); return; }
        // planes worth computing: the owned ones and out_valid ghost planes per side
        const bool can_window = overlap_exchanges && fuse_housekeeping && fl_get_option(FL_OPT_FUSED_HOUSEKEEPING) >= 0;
        const int ov = out_valid < 0 ? 0 : (out_valid > slab.G ? slab.G : out_valid);
        // (at least one plane on the high side: the last rank owns the w face on top of its last cell plane)
        const int w0 = can_window ? slab.G - ov : 0, w1 = can_window ? g.nk - slab.G + (ov > 1 ? ov : 1) : g.nk;
        float *ptrs[16]; size_t planes[16]; int extras[16]; DeviceField *moved[16];
        int n = 0, reach = 0;
        for (const GhostNeed &nd : needs) {
            if (nd.depth > slab.G) { require(nd.fields, nd.depth); return; }     // latches the "ghost zone too shallow" error
            for (DeviceField *f : nd.fields) {
                if (f->valid >= nd.depth) continue;
                bool seen = false;
                for (int a = 0; a < n; a++) seen = seen || moved[a] == f;
                if (seen || n >= 16) continue;
                moved[n] = f; ptrs[n] = f->get(); planes[n] = f->plane; extras[n] = f->extra; n++;
                if (nd.depth > reach) reach = nd.depth;
            }
        }
        const bool trimmed = can_window && (w0 > 0 || w1 < g.nk);
        if (!n) {
            if (trimmed && fl_set_plane_window(w0, w1) == 1) { op(); fl_set_plane_window(-1, -1); }
            else op();
            return;
        }
        bool hazard = false;
        for (const DeviceField *wf : writes)
            for (int a = 0; a < n; a++) hazard = hazard || wf == moved[a] || wf->get() == ptrs[a];
        if (hazard) fl_report_error(FL_ERR_BAD_ARGUMENT, "withGhosts: the operator writes a field whose ghost planes are in flight");
        const int k0 = slab.G + reach, k1 = g.nk - slab.G - reach;
        const bool split = !hazard && can_window && k1 - k0 >= 8 && fl_set_plane_window(k0, k1) == 1;
        if (!split) {
            for (const GhostNeed &nd : needs) require(nd.fields, nd.depth);
            if (trimmed && fl_set_plane_window(w0, w1) == 1) { op(); fl_set_plane_window(-1, -1); }
            else op();
            return;
        }
        if (trace_require() && slab.rank == 0)
            fprintf(stderr, "[require] %d fields, reach %d: exchange overlapped with planes [%d, %d) of %d\n", n, reach, k0, k1, g.nk);
        // (BQ_OPT_SHALLOW_BLOCKING_EXCHANGE = 2: only the planes the operator can reach travel)
        const int depth_moved = shallow_blocking >= 2 ? reach : slab.G;
        fl_halo_exchange(n, ptrs, planes, extras, g.nk, slab.G, depth_moved, /*wait=*/0);
        op();                                       // planes [k0, k1): no ghost plane within reach
        fl_halo_wait();
        for (int a = 0; a < n; a++) moved[a]->valid = depth_moved;
        fl_set_plane_window(w0, k0); op();
        fl_set_plane_window(k1, w1); op();
        fl_set_plane_window(-1, -1);
    }
    // z-slab ranks, reference-faithful DMC border (wall_sheets.hpp): re-evaluate the wall layers of stage 3 of the
    // compensation (dst = before + blend9(coeff * src(psi_back(x)))) from a copy of `src` assembled out of the sheets
    // the border taps can touch, fetched from the ranks that own them.  Call right after the stage-3 operator, with
    // the out_valid that operator was given.  `need`: ghost planes of src known to be correct (the operator's
    // requirement).  No-op on one rank.
    std::vector<std::pair<const DeviceField *, const DeviceField *>> global_twins_;
    struct WallItem { DeviceField *src, *before, *dst; FieldKind kind; };
    void wallFixup(std::initializer_list<WallItem> items, DeviceField &bx, DeviceField &by, DeviceField &bz,
                   int Dback, int need, float coeff, int out_valid);
    // the same in two halves around the stage-3 operator: Begin gathers what the other ranks need and starts the
    // messages (they travel while the operator runs), End places what arrived and re-evaluates the wall layers
    void wallFixupBegin(std::initializer_list<WallItem> items, int Dback, int need);
    void wallFixupEnd(DeviceField &bx, DeviceField &by, DeviceField &bz, float coeff, int out_valid);
    long long wall_bytes_moved = 0;         // floats received through wallFixup so far x 4 (statistics)
    // blend != 1 on z-slab ranks in the reference-faithful mode: whole-grid copies of the *Prev fields (include/bimocq_gpu.h:
    // gpu_advect_vel_double_global), assembled from every rank's owned planes after a re-initialisation -- the only time those
    // fields change.  globalTwin: the copy registered for a local field, or nullptr (local look-ups).
    struct GlobalPair { DeviceField *local, *global; };
    bool assembleGlobal(std::initializer_list<GlobalPair> fields);
    const float *globalTwin(const DeviceField *local) const
    {
        for (const std::pair<const DeviceField *, const DeviceField *> &t : global_twins_)
            if (t.first == local) return t.second->get();
        return nullptr;
    }
    void dropGlobalTwins() { global_twins_.clear(); }
    long long global_prev_bytes = 0;        // bytes received by assembleGlobal so far (statistics)
    bool overlap_exchanges = true;          // BQ_OPT_OVERLAP_EXCHANGES
    bool jacobi_ends_first = true;          // BQ_OPT_JACOBI_ENDS_FIRST
    bool jacobi_triples = true;             // BQ_OPT_JACOBI_TRIPLES
    bool concurrent_maps = false;           // BQ_OPT_CONCURRENT_MAPS (measured: no gain, EXPERIMENTS.md section 9)
    int shallow_blocking = 0;               // BQ_OPT_SHALLOW_BLOCKING_EXCHANGE: 1 require(), 2 also withGhosts() move only the planes asked for
    static bool trace_require() { static const bool on = getenv("BQ_TRACE_REQUIRE") && atoi(getenv("BQ_TRACE_REQUIRE")) != 0; return on; }
    // record that an operator just rewrote `f` from inputs whose reach left `valid` correct ghost planes
    void produced(DeviceField &f, int valid) const { if (slab.on) f.valid = valid < 0 ? 0 : valid; }
    void producedAll(std::initializer_list<DeviceField *> fields, int valid) const { for (DeviceField *f : fields) produced(*f, valid); }
    // the smallest ghost validity among `fields` once require(fields, depth) has been met (what an exchange refreshes
    // comes back with all G planes)
    int validAfter(std::initializer_list<const DeviceField *> fields, int depth) const
    {
        int v = DeviceField::kAlwaysValid;
        for (const DeviceField *f : fields) { const int a = (slab.on && slab.nranks > 1 && f->valid < depth) ? slab.G : f->valid; if (a < v) v = a; }
        return v;
    }
    static int minValid(std::initializer_list<const DeviceField *> fields)
    {
        int v = DeviceField::kAlwaysValid;
        for (const DeviceField *f : fields) if (f->valid < v) v = f->valid;
        return v;
    }
    // scratch owned by the mapper (GPU_Advection.h:122-136)
    DeviceField u_src, v_src, w_src;
    bool fuse_housekeeping = true;          // BQ_OPT_FUSED_HOUSEKEEPING (see FusedScope)
    DeviceField x_out, y_out, z_out;        // DMC ping buffers (border nodes stay 0, as in the reference)
    DeviceField x_out2, y_out2, z_out2;     // second ping set so sub-steps never copy

    // event pair of startEventRecord/endEventRecord (GPU_Advection.h:228-247)
    void startEventRecord();
    float endEventRecord();

    void solveForward(float *u, float *v, float *w, float *xf, float *yf, float *zf, float cfldt, float dt) const
    { gpu_solve_forward(u, v, w, xf, yf, zf, g.h, g.ni, g.nj, g.nk, cfldt, dt); }

    // one DMC sub-step in -> out (GPU_Advection.h:460-470 without the copy-back; the caller swaps)
    void solveBackwardDMC(float *u, float *v, float *w, float *xi, float *yi, float *zi,
                          float *xo, float *yo, float *zo, float substep) const
    { gpu_solve_backwardDMC(u, v, w, xi, yi, zi, xo, yo, zo, g.h, g.ni, g.nj, g.nk, substep); }

    void advectVelocity(float *u, float *v, float *w, float *ui, float *vi, float *wi,
                        float *bx, float *by, float *bz, bool is_point) const;
    void advectVelocityDouble(float *u, float *v, float *w, float *ut, float *vt, float *wt,
                              float *bx, float *by, float *bz, float *px, float *py, float *pz,
                              bool is_point, float blend) const
    { gpu_advect_vel_double(u, v, w, ut, vt, wt, bx, by, bz, px, py, pz, g.h, g.ni, g.nj, g.nk, is_point, blend); }
    void advectVelocityDoubleGlobal(float *u, float *v, float *w, const float *ug, const float *vg, const float *wg,
                                    float *bx, float *by, float *bz, float *px, float *py, float *pz,
                                    bool is_point, float blend) const
    {
        gpu_advect_vel_double_global(u, v, w, const_cast<float *>(ug), const_cast<float *>(vg), const_cast<float *>(wg),
                                     bx, by, bz, px, py, pz, g.h, g.ni, g.nj, g.nk, is_point, blend);
    }
    void compensateVelocity(float *u, float *v, float *w, float *du, float *dv, float *dw,
                            float *fx, float *fy, float *fz, float *bx, float *by, float *bz, bool is_point) const;
    void advectField(float *f, float *fi, float *bx, float *by, float *bz, bool is_point) const;
    void advectFieldDouble(float *f, float *fp, float *bx, float *by, float *bz,
                           float *px, float *py, float *pz, bool is_point, float blend) const
    { gpu_advect_field_double(f, fp, bx, by, bz, px, py, pz, g.h, g.ni, g.nj, g.nk, is_point, blend); }
    void advectFieldDoubleGlobal(float *f, const float *fg, float *bx, float *by, float *bz,
                                 float *px, float *py, float *pz, bool is_point, float blend) const
    { gpu_advect_field_double_global(f, const_cast<float *>(fg), bx, by, bz, px, py, pz, g.h, g.ni, g.nj, g.nk, is_point, blend); }
    void compensateField(float *f, float *df, float *fx, float *fy, float *fz,
                         float *bx, float *by, float *bz, bool is_point) const;
    void accumulateVelocity(float *uc, float *vc, float *wc, float *dui, float *dvi, float *dwi,
                            float *fx, float *fy, float *fz, bool is_point, float coeff) const
    { gpu_accumulate_velocity(uc, vc, wc, dui, dvi, dwi, fx, fy, fz, g.h, g.ni, g.nj, g.nk, is_point, coeff); }
    void accumulateField(float *fc, float *dfi, float *fx, float *fy, float *fz, bool is_point, float coeff) const
    { gpu_accumulate_field(fc, dfi, fx, fy, fz, g.h, g.ni, g.nj, g.nk, is_point, coeff); }
    void emitSmoke(float *u, float *v, float *w, float *rho, float *T, float cx, float cy, float cz,
                   float radius, float density, float temperature, float emiter) const
    { gpu_emit_smoke(u, v, w, rho, T, g.h, g.ni, g.nj, g.nk, cx, cy, cz, radius, density, temperature, emiter); }
    void add_buoyancy(float *v, float *rho, float *T, float alpha, float beta, float dt) const
    { gpu_add_buoyancy(v, rho, T, g.ni, g.nj, g.nk, alpha, beta, dt); }
    void diffuseField(float *f, float *t0, float *t1, int ni, int nj, int nk, int iter, float coef) const
    { gpu_diffuse_field(f, t0, t1, ni, nj, nk, iter, coef); }
    void addFields(float *out, float *f1, float *f2, float coeff, size_t number) const
    { gpu_add_field(out, f1, f2, coeff, (int)number); }
    void add(float *f1, float *f2, float coeff, size_t number) const { gpu_add(f1, f2, coeff, (int)number); }
    void projectionJacobi(float *u, float *v, float *w, float *div, float *p, float *p_temp, float *debug,
                          int iter, float halfrdx, float alpha, float beta) const;

private:
    bool ok_ = false;
    void *ev_start_ = nullptr, *ev_stop_ = nullptr;
    // wall sheets: plans per (field kind, Dback, need); assembled copies (one per item of a batch), staging buffers
    std::map<std::tuple<int, int, int>, WallPlan> wall_plans_;
    struct Shadow { DeviceField buf; int k0 = 0, k1 = 0; size_t plane = 0; };
    Shadow wall_shadow_[3];
    DeviceField wall_send_, wall_recv_;
    // between wallFixupBegin and wallFixupEnd
    int wall_n_ = 0;
    WallItem wall_items_[3] = {};
    const WallPlan *wall_plans_now_[3] = { nullptr, nullptr, nullptr };
    size_t wall_send_base_[3] = { 0, 0, 0 }, wall_recv_base_[3] = { 0, 0, 0 };
    void wallDims(FieldKind kind, int &nbi, int &nbj, int &nkf) const
    {
        nbi = g.ni + (kind == FIELD_U); nbj = g.nj + (kind == FIELD_V); nkf = g.nk + (kind == FIELD_W);
    }
    const WallPlan &wallPlan(FieldKind kind, int Dback, int need);
};

} // namespace bqhost
