// forma_host.h — C++ host mirror of the part of forma's public API that feeds the hot path
// (reference forma/src/lib.rs:130-154): Point, PathBuilder, Path, plus the batching of stage-1 work
// items for the HIP flatten kernel.  Same method names and argument meaning as the reference.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "../../include/forma_hip.h"

namespace forma {

struct Point { float x, y; };                                  // math/point.rs:23-27

// Work items of one path for the flatten kernel (the output of the sequential half of
// `Primitives`, path.rs:252-445); indices are path-relative.
struct FlattenPlan {
    std::vector<uint32_t> point_commands, point_indices, quad_indices;
    std::vector<uint8_t>  new_contour;                         // start_new_contour, path.rs:567-572
    std::vector<float> qx, qy, qw, x0, dx_recip, k0, dk, curvatures_recip, partial_curv;
    std::vector<uint32_t> partial_spline;
    std::vector<float> sp0x, sp0y, sp2x, sp2y;
    struct Walker;
};

class Path {                                                   // path.rs:669-771
public:
    struct Data;
    Path transform(const float t[9]) const;
    const FlattenPlan& plan() const;                           // memoised, like PathData::segments
    bool has_affine() const { return has_affine_; }
    const float* affine() const { return affine_; }            // ux uy vx vy tx ty
private:
    friend class PathBuilder;
    std::shared_ptr<Data> d_;
    bool has_affine_ = false;
    float affine_[6] = {1, 0, 0, 1, 0, 0};
};

class PathBuilder {                                            // path.rs:773-925
public:
    PathBuilder();
    PathBuilder& move_to(Point p);
    PathBuilder& line_to(Point p);
    PathBuilder& quad_to(Point p1, Point p2);
    PathBuilder& cubic_to(Point p1, Point p2, Point p3);
    PathBuilder& rat_quad_to(Point p1, Point p2, float weight);
    PathBuilder& rat_cubic_to(Point p1, Point p2, Point p3, float w1, float w2);
    Path build();
private:
    std::shared_ptr<Path::Data> d_;
};

// Concatenated work items of many paths: one k_flatten launch for a whole composition.
struct FlattenBatch : FlattenPlan {
    std::vector<uint32_t> line_slot;                           // per output point (reference `ids`)
    struct Affine { size_t first, count; float m[6]; };
    std::vector<Affine> affines;
    void add(const Path& path, uint32_t slot);
    void tables(forma_flatten_tables_t* t) const;
    void apply_affines(float* x, float* y) const;
};

}  // namespace forma
