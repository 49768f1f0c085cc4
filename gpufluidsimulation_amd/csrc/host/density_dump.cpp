// density_dump.cpp -- the per-frame density dump behind outputResult().
//
// Contract restated from the reference's writeVDB (src/utils/volumeMeshTools.h:33-60): visit voxels
// in k, j, i order; keep |value| where it exceeds 1e-4 (a float compared against the double literal);
// one grid named "density", class fog volume, linear transform = voxel size; file
// <path>/density_render_%04d.<ext> numbered frame (the caller passes frame + 1).
// OpenVDB is not available in this image, so the default container is a dependency-free
// sparse format ("BQDENS01", little endian) that carries exactly the same information; a z-slab rank
// writes its own planes with k_offset so that eight ranks' files concatenate to the global grid.
// Built with -DHAVE_OPENVDB (make HAVE_OPENVDB=1; needs the OpenVDB headers and library, never built in this image --
// tests/test_host_logic_cpu.py type-checks the branch against a declaration-only model of the API, tests/vdb_decl) the
// same voxels go into a real .vdb as well (write_density_vdb below).
#include "fluid_solver.hpp"

#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <sys/stat.h>
#ifdef HAVE_OPENVDB
#include <openvdb/openvdb.h>
#endif

namespace bqhost {

namespace {
#pragma pack(push, 1)
struct DumpHeader {
    char     magic[8];          // "BQDENS01"
    uint32_t frame;
    int32_t  nx, ny, nz;        // GLOBAL grid dims
    int32_t  k_offset, nz_local;// planes [k_offset, k_offset + nz_local) are in this file
    float    voxel_size;        // Transform::createLinearTransform(voxel_size)
    float    threshold;         // 1e-4
    char     grid_name[16];     // "density"
    uint32_t grid_class;        // 1 = GRID_FOG_VOLUME
    uint64_t count;             // records that follow
};
struct DumpRecord { int32_t i, j, k; float value; };
#pragma pack(pop)

bool make_dirs(const std::string &path)
{
    std::string cur;
    for (size_t a = 0; a <= path.size(); a++) {
        if (a == path.size() || path[a] == '/') {
            if (!cur.empty() && cur != "." && cur != "..") {
                if (mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return false;
            }
        }
        if (a < path.size()) cur.push_back(path[a]);
    }
    return true;
}
} // namespace

long write_density_dump(unsigned frame, const std::string &filepath, float voxel_size,
                        const float *density, int nx, int ny, int nz, int k_offset, int nz_global)
{
    if (!make_dirs(filepath)) return -1;       // boost::filesystem::create_directories in main.cpp:138
    char name[512];
    if (k_offset == 0 && nz == nz_global)
        snprintf(name, sizeof name, "%s/density_render_%04u.bqd", filepath.c_str(), frame);
    else
        snprintf(name, sizeof name, "%s/density_render_%04u.k%05d.bqd", filepath.c_str(), frame, k_offset);
    FILE *f = fopen(name, "wb");
    if (!f) return -1;
    DumpHeader hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, "BQDENS01", 8);
    hd.frame = frame; hd.nx = nx; hd.ny = ny; hd.nz = nz_global; hd.k_offset = k_offset; hd.nz_local = nz;
    hd.voxel_size = voxel_size; hd.threshold = 1e-4f;
    strncpy(hd.grid_name, "density", sizeof hd.grid_name - 1);
    hd.grid_class = 1;
    if (fwrite(&hd, sizeof hd, 1, f) != 1) { fclose(f); return -1; }
    uint64_t total = 0;
    DumpRecord buf[4096];
    size_t nb = 0;
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                float value = std::fabs(density[(size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k)]);
                if ((double)value > 1e-4) {
                    buf[nb++] = DumpRecord{ i, j, k + k_offset, value };
                    total++;
                    if (nb == 4096) { if (fwrite(buf, sizeof(DumpRecord), nb, f) != nb) { fclose(f); return -1; } nb = 0; }
                }
            }
    if (nb && fwrite(buf, sizeof(DumpRecord), nb, f) != nb) { fclose(f); return -1; }
    hd.count = total;
    if (fseek(f, 0, SEEK_SET) != 0 || fwrite(&hd, sizeof hd, 1, f) != 1) { fclose(f); return -1; }
    fclose(f);
#ifdef HAVE_OPENVDB
    if (write_density_vdb(frame, filepath, voxel_size, density, nx, ny, nz, k_offset, nz_global) < 0) return -1;
#endif
    return (long)total;
}

#ifdef HAVE_OPENVDB
// The real thing, for boxes that have OpenVDB: <path>/density_render_%04d.vdb (slab ranks:
// .k%05d.vdb) holding one float grid "density" -- fog volume, linear transform of the voxel size, active
// voxels = the cells whose |rho| exceeds 1e-4, stored value |rho|, global (i, j, k) coordinates.
long write_density_vdb(unsigned frame, const std::string &filepath, float voxel_size,
                       const float *density, int nx, int ny, int nz, int k_offset, int nz_global)
{
    static const bool initialised = (openvdb::initialize(), true);
    (void)initialised;
    auto grid = openvdb::FloatGrid::create();
    grid->setName("density");
    grid->setGridClass(openvdb::GRID_FOG_VOLUME);
    grid->setTransform(openvdb::math::Transform::createLinearTransform(voxel_size));
    auto acc = grid->getAccessor();
    long total = 0;
    const size_t plane = (size_t)nx * ny;
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++) {
            const float *row = density + plane * k + (size_t)nx * j;
            for (int i = 0; i < nx; i++) {
                const float a = std::fabs(row[i]);
                if ((double)a > 1e-4) { acc.setValue(openvdb::Coord(i, j, k + k_offset), a); total++; }
            }
        }
    char name[512];
    if (k_offset == 0 && nz == nz_global)
        snprintf(name, sizeof name, "%s/density_render_%04u.vdb", filepath.c_str(), frame);
    else
        snprintf(name, sizeof name, "%s/density_render_%04u.k%05d.vdb", filepath.c_str(), frame, k_offset);
    try {
        openvdb::io::File out(name);
        openvdb::GridPtrVec grids;              // (io::File::write is a template over the container: a braced list would not deduce)
        grids.push_back(grid);
        out.write(grids);
        out.close();
    } catch (const std::exception &) {
        return -1;
    }
    return total;
}
#endif

} // namespace bqhost
