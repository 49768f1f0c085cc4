// TEST INFRASTRUCTURE ONLY -- a declaration-only model of the slice of OpenVDB's public interface that
// csrc/host/density_dump.cpp uses behind HAVE_OPENVDB, so that `g++ -fsyntax-only -DHAVE_OPENVDB` can type-check that
// branch in an image without OpenVDB (tests/test_host_logic_cpu.py::test_vdb_branch_type_checks).  Nothing here is ever
// linked or shipped, and it is no substitute for building against the real library: it only keeps typos, missing
// includes and calls of the wrong shape out of a branch no compiler in this image would otherwise read.
//
// Shapes follow OpenVDB's documented API (openvdb/openvdb.h, Grid.h, io/File.h, math/Transform.h, math/Coord.h):
//   * FloatGrid::create() returns FloatGrid::Ptr (a shared_ptr); Grid derives from GridBase
//   * io::File::write is a TEMPLATE over the grid-pointer container -- a braced list does not deduce; callers pass a
//     GridPtrVec / GridCPtrVec object (as the reference's writeVDB does, src/utils/volumeMeshTools.h:56-58)
//   * Transform::createLinearTransform(double voxelSize) returns Transform::Ptr
#pragma once
#include <memory>
#include <string>
#include <vector>

namespace openvdb {

void initialize();

enum GridClass { GRID_UNKNOWN = 0, GRID_LEVEL_SET, GRID_FOG_VOLUME, GRID_STAGGERED };

namespace math {
class Coord {
public:
    Coord();
    Coord(int x, int y, int z);
};
class Transform {
public:
    using Ptr = std::shared_ptr<Transform>;
    static Ptr createLinearTransform(double voxelSize = 1.0);
};
} // namespace math
using math::Coord;

class MetaMap {};

class GridBase {
public:
    using Ptr = std::shared_ptr<GridBase>;
    using ConstPtr = std::shared_ptr<const GridBase>;
    virtual ~GridBase();
    void setName(const std::string &);
    void setGridClass(GridClass);
    void setTransform(math::Transform::Ptr);
};
using GridPtrVec = std::vector<GridBase::Ptr>;
using GridCPtrVec = std::vector<GridBase::ConstPtr>;

template <typename ValueT>
class ValueAccessorModel {
public:
    void setValue(const Coord &xyz, const ValueT &value);
};

template <typename ValueT>
class GridModel : public GridBase {
public:
    using Ptr = std::shared_ptr<GridModel>;
    using Accessor = ValueAccessorModel<ValueT>;
    static Ptr create();
    Accessor getAccessor();
};
using FloatGrid = GridModel<float>;

namespace io {
class File {
public:
    explicit File(const std::string &filename);
    ~File();
    template <typename GridPtrContainerT>
    void write(const GridPtrContainerT &, const MetaMap & = MetaMap()) const;
    void close();
};
} // namespace io

} // namespace openvdb
