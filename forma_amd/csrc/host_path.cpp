// host_path.cpp — host half of stage 1 (C++ mirror of forma's `PathBuilder` / `Path`, reference
// forma/src/path.rs:776-925, and of the SEQUENTIAL part of curve flattening, path.rs:206-445).
//
// The reference splits flattening in two: `Primitives::push_{line,quad,cubic}` + `populate_buffers`
// walk the path once, carrying spline/angle/curvature state from one primitive to the next (inherently
// sequential), and emit one small work item per output point; `into_segments` then evaluates every
// point independently (path.rs:487-534).  This file is the first half; the second half is the HIP
// kernel k_flatten (lines.hip).  The work items of many paths are concatenated and flattened in a
// single launch.  Product code: no dependency on oracle/.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/forma_hip.h"
#include "forma_host.h"

namespace forma {

namespace {
constexpr float kMaxError = 1.0f / 16.0f;        // path.rs:40  (half a sub-pixel)
constexpr float kMaxAngleError = 0.001f;         // path.rs:41
constexpr float kEps = 1.1920929e-7f;            // f32::EPSILON
constexpr float kPi = 3.14159265358979323846f;
constexpr float kHalfPi = 1.57079632679489661923f;

struct V2 { float x, y; };
inline V2 sub(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline float norm(V2 p) { return std::sqrt(p.x * p.x + p.y * p.y); }

// math/point.rs:53-90: polynomial atan2; an angle exists only for vectors longer than EPSILON
inline bool direction(V2 d, float* angle) {
    if (!(norm(d) >= kEps)) return false;
    float xa = std::fabs(d.x), ya = std::fabs(d.y);
    float a = fminf(xa, ya) / fmaxf(xa, ya);
    float s = a * a;
    float r = fmaf(fmaf(fmaf(s, -0.046496473f, 0.15931422f), s, -0.32762277f), s * a, a);
    if (ya > xa) r = kHalfPi - r;
    if (d.x < 0.0f) r = kPi - r;
    if (d.y < 0.0f) r = -r;
    *angle = r;
    return true;
}
inline float mix(float t, float a, float b) { return fmaf(t, b, fmaf(-t, a, a)); }     // lerp, path.rs:44-46
inline float curvature_map(float x) {                                                   // path.rs:48-51
    const float c = 0.67f;
    return x / (1.0f - c + std::sqrt(std::sqrt(fmaf(x * x, 0.25f, c * c * c * c))));
}
struct HP { float x, y, w; };                                                           // WeightedPoint
inline V2 project(HP p) { float r = 1.0f / p.w; return {p.x * r, p.y * r}; }            // applied(), path.rs:64-73
inline float bez3(float t, float a, float b, float c, float d) {
    return mix(t, mix(t, mix(t, a, b), mix(t, b, c)), mix(t, mix(t, b, c), mix(t, c, d)));
}
inline HP cubic_at(float t, const HP* p) {                                              // eval_cubic, path.rs:75-120
    return {bez3(t, p[0].x, p[1].x, p[2].x, p[3].x), bez3(t, p[0].y, p[1].y, p[2].y, p[3].y),
            bez3(t, p[0].w, p[1].w, p[2].w, p[3].w)};
}
inline size_t to_count(float v) { return (v > 0.0f) ? (v >= 1.8446744e19f ? (size_t)-1 : (size_t)v) : 0; }
}  // namespace

// ---- the sequential curvature walk ---------------------------------------------------------------
struct FlattenPlan::Walker {
    FlattenPlan& out;
    bool pending_contour = true;         // Primitives::contour (Default: Some, path.rs:541-558)
    bool have_angle = false; float last_angle = 0;
    struct Sp { float curvature; V2 p0, p2; bool owns_contour; };
    std::vector<Sp> splines;
    std::vector<float> partial_total;    // partial_curvatures[i].1
    std::vector<uint32_t> partial_sp;    // partial_curvatures[i].0

    explicit Walker(FlattenPlan& o) : out(o) {}

    // last_spline_or_insert_with, path.rs:206-246
    Sp& spline_for(bool has_angle, float angle, V2 start, V2 end_hint) {
        bool fresh = false;
        if (pending_contour) { pending_contour = false; fresh = true; }
        else {
            bool turned = false;
            if (have_angle && has_angle) {
                float d = std::fabs(angle - last_angle);
                if (d > kPi) d -= kPi;
                if (d > kHalfPi) d = kPi - d;
                turned = d > kMaxAngleError;
            }
            if (!splines.empty()) {
                Sp& last = splines.back();
                if ((turned || norm(sub(start, last.p2)) >= kMaxError) && last.owns_contour) {
                    last.owns_contour = false;                 // the contour token moves to the new spline
                    fresh = true;
                }
            }
        }
        if (fresh) splines.push_back({0.0f, start, end_hint, true});
        return splines.back();
    }

    void line(HP a, HP b) {                                    // push_line, path.rs:252-269
        V2 p0 = project(a), p1 = project(b);
        float ang = 0; bool has = direction(sub(p1, p0), &ang);
        Sp& s = spline_for(has, ang, p0, p1);
        s.p2 = p1;
        have_angle = has; last_angle = ang;
    }

    void quad(HP q0, HP q1, HP q2) {                           // push_quad, path.rs:271-347
        V2 p0 = project(q0), p1 = project(q1), p2 = project(q2);
        V2 a = sub(p1, p0), b = sub(p2, p1);
        float ain = 0, aout = 0;
        bool hin = direction(a, &ain), hout = direction(b, &aout);
        if (!hin && !hout) return;
        if (!hin || !hout) { line(q0, q2); return; }
        for (HP q : {q0, q1, q2}) { out.qx.push_back(q.x); out.qy.push_back(q.y); out.qw.push_back(q.w); }
        Sp& s = spline_for(true, ain, p0, p2);
        s.p2 = p2;
        V2 h = sub(a, b);
        float cross = fmaf(p2.x - p0.x, h.y, -(p2.y - p0.y) * h.x);
        float cross_r = 1.0f / cross;
        float x0 = fmaf(a.x, h.x, a.y * h.y) * cross_r;
        float x2 = fmaf(b.x, h.x, b.y * h.y) * cross_r;
        float dxr = 1.0f / (x2 - x0);
        float scale = std::fabs(cross / (norm(h) * (x2 - x0)));
        float k0 = curvature_map(x0), k2 = curvature_map(x2);
        float dk = k2 - k0;
        float cur = 0.5f * std::fabs(dk) * std::sqrt(scale * (1.0f / kMaxError));
        if (!std::isfinite(cur) || cur <= 1.0f) {              // collinear control points
            x0 = 0.03662467f; dxr = 1.0f; k0 = 0.0f; dk = 1.0f; cur = 2.0f;
        }
        float total = s.curvature + cur;
        s.curvature = total;
        have_angle = true; last_angle = aout;
        out.x0.push_back(x0); out.dx_recip.push_back(dxr); out.k0.push_back(k0); out.dk.push_back(dk);
        out.curvatures_recip.push_back(1.0f / cur);
        partial_sp.push_back((uint32_t)splines.size() - 1); partial_total.push_back(total);
    }

    void cubic(const HP* q) {                                  // push_cubic, path.rs:349-398
        const float bound = (36.0f * 36.0f / 3.0f) * kMaxError * kMaxError;
        V2 p0 = project(q[0]), p1 = project(q[1]), p2 = project(q[2]);
        float dx = fmaf(p2.x, 3.0f, -p0.x) - fmaf(p1.x, 3.0f, -p1.x);
        float dy = fmaf(p2.y, 3.0f, -p0.y) - fmaf(p1.y, 3.0f, -p1.y);
        float err = fmaf(dx, dx, dy * dy);
        float mult = fmaxf(fmaxf(q[1].w, q[2].w), 1.0f);
        size_t n = to_count(std::ceil(powf(err * (1.0f / bound), 1.0f / 6.0f) * mult));
        if (n < 1) n = 1;
        float step = 1.0f / (float)n;
        V2 from = p0;
        for (size_t i = 1; i <= n; i++) {
            float t = (float)i * step;
            V2 to = project(cubic_at(t, q));
            V2 mid = project(cubic_at(t - 0.5f * step, q));
            V2 ctrl = {fmaf(mid.x, 2.0f, -0.5f * (from.x + to.x)), fmaf(mid.y, 2.0f, -0.5f * (from.y + to.y))};
            quad({from.x, from.y, 1.0f}, {ctrl.x, ctrl.y, 1.0f}, {to.x, to.y, 1.0f});
            from = to;
        }
    }

    // populate_buffers, path.rs:400-445: one work item per output point
    void emit() {
        size_t qi = 0;
        const Sp* prev = nullptr;
        for (size_t si = 0; si < splines.size(); si++) {
            const Sp& sp = splines[si];
            size_t n = to_count(std::ceil(sp.curvature));
            float incr = sp.curvature / (float)n;
            uint32_t incr_bits; std::memcpy(&incr_bits, &incr, 4);
            if (!prev || prev->owns_contour || norm(sub(prev->p2, sp.p0)) > kMaxError) {
                out.point_commands.push_back(0x7F800000u | ((uint32_t)si & 0x3FFFFFu));          // Start
                out.point_indices.push_back(0); out.quad_indices.push_back(0); out.new_contour.push_back(0);
            }
            for (size_t pi = 1; pi < n; pi++) {
                if ((float)pi > partial_total[qi]) qi++;
                out.point_commands.push_back(incr_bits);                                          // Incr
                out.point_indices.push_back((uint32_t)pi); out.quad_indices.push_back((uint32_t)qi);
                out.new_contour.push_back(0);
            }
            out.point_commands.push_back(0xFF800000u | ((uint32_t)si & 0x3FFFFFu) | (sp.owns_contour ? 1u << 22 : 0u));  // End
            out.point_indices.push_back(0); out.quad_indices.push_back(0);
            out.new_contour.push_back(sp.owns_contour ? 1 : 0);
            prev = &sp;
            if (n > 0) qi++;
        }
        for (const Sp& sp : splines) { out.sp0x.push_back(sp.p0.x); out.sp0y.push_back(sp.p0.y); out.sp2x.push_back(sp.p2.x); out.sp2y.push_back(sp.p2.y); }
        out.partial_spline = partial_sp; out.partial_curv = partial_total;
    }
};

// ---- PathData / PathBuilder / Path (path.rs:574-925) ------------------------------------------------
struct Path::Data {
    std::vector<float> x{0.0f}, y{0.0f}, w{1.0f};
    std::vector<uint8_t> cmd{0};            // 0 Move, 1 Line, 2 Quad, 3 Cubic
    size_t open_index = 0;
    std::unique_ptr<FlattenPlan> plan;      // memoised `segments` (path.rs:617-654)

    void close() {                          // path.rs:596-615
        size_t n = x.size();
        V2 a = project({x[n - 1], y[n - 1], w[n - 1]});
        V2 b = project({x[open_index], y[open_index], w[open_index]});
        if (!(a.x == b.x && a.y == b.y)) {
            x.push_back(x[open_index]); y.push_back(y[open_index]); w.push_back(w[open_index]);
            cmd.push_back(1);
        }
    }
    void add(float px, float py, float pw) { x.push_back(px); y.push_back(py); w.push_back(pw); }
};

PathBuilder::PathBuilder() : d_(std::make_shared<Path::Data>()) {}
PathBuilder& PathBuilder::move_to(Point p) {                    // path.rs:783-810
    Path::Data& d = *d_;
    size_t n = d.x.size();
    if (d.cmd.back() == 0) { d.x[n - 1] = p.x; d.y[n - 1] = p.y; d.w[n - 1] = 1.0f; }
    else {
        d.close();
        size_t open = d.x.size();
        d.add(p.x, p.y, 1.0f); d.cmd.push_back(0); d.open_index = open;
    }
    return *this;
}
PathBuilder& PathBuilder::line_to(Point p) { d_->add(p.x, p.y, 1.0f); d_->cmd.push_back(1); return *this; }
PathBuilder& PathBuilder::quad_to(Point p1, Point p2) { d_->add(p1.x, p1.y, 1); d_->add(p2.x, p2.y, 1); d_->cmd.push_back(2); return *this; }
PathBuilder& PathBuilder::cubic_to(Point p1, Point p2, Point p3) {
    d_->add(p1.x, p1.y, 1); d_->add(p2.x, p2.y, 1); d_->add(p3.x, p3.y, 1); d_->cmd.push_back(3); return *this;
}
PathBuilder& PathBuilder::rat_quad_to(Point p1, Point p2, float weight) {   // control point stored pre-multiplied, :872-889
    d_->add(p1.x * weight, p1.y * weight, weight); d_->add(p2.x, p2.y, 1); d_->cmd.push_back(2); return *this;
}
PathBuilder& PathBuilder::rat_cubic_to(Point p1, Point p2, Point p3, float w1, float w2) {
    d_->add(p1.x * w1, p1.y * w1, w1); d_->add(p2.x * w2, p2.y * w2, w2); d_->add(p3.x, p3.y, 1); d_->cmd.push_back(3);
    return *this;
}
Path PathBuilder::build() { d_->close(); Path p; p.d_ = d_; return p; }   // shares the data like the reference (Rc)

const FlattenPlan& Path::plan() const {
    Data& d = *d_;
    if (!d.plan) {
        d.plan = std::make_unique<FlattenPlan>();
        FlattenPlan::Walker wk(*d.plan);
        size_t i = 0;
        auto P = [&](size_t k) { return HP{d.x[k], d.y[k], d.w[k]}; };
        for (uint8_t c : d.cmd) {
            switch (c) {
                case 0: i += 1; wk.pending_contour = true; break;
                case 1: i += 1; wk.line(P(i - 2), P(i - 1)); break;
                case 2: i += 2; wk.quad(P(i - 3), P(i - 2), P(i - 1)); break;
                default: { i += 3; HP q[4] = {P(i - 4), P(i - 3), P(i - 2), P(i - 1)}; wk.cubic(q); }
            }
        }
        wk.emit();
    }
    return *d.plan;
}

// Path::transform, path.rs:725-771: an affine matrix that does not scale up stays a cheap per-point
// transform (GeomPresTransform, math/transform.rs:151-222); anything else re-transforms the control points.
Path Path::transform(const float t[9]) const {
    const float eps = kEps;
    if (std::fabs(t[6]) <= eps && std::fabs(t[7]) <= eps) {
        float m[6] = {t[0], t[1], t[2], t[3], t[4], t[5]};
        if (std::fabs(t[8] - 1.0f) > eps) { float r = 1.0f / t[8]; for (float& v : m) v *= r; }
        float ux = m[0], vx = m[1], tx = m[2], uy = m[3], vy = m[4], ty = m[5];
        const float max_x = 1.0f + kMaxError / 65536.0f, max_y = 1.0f + kMaxError / 32768.0f;
        if (!(ux * ux + uy * uy > max_x) && !(vx * vx + vy * vy > max_y)) {
            Path p; p.d_ = d_; p.has_affine_ = true;
            p.affine_[0] = ux; p.affine_[1] = uy; p.affine_[2] = vx; p.affine_[3] = vy; p.affine_[4] = tx; p.affine_[5] = ty;
            return p;
        }
    }
    Path p;
    p.d_ = std::make_shared<Data>();
    p.d_->x = d_->x; p.d_->y = d_->y; p.d_->w = d_->w; p.d_->cmd = d_->cmd; p.d_->open_index = d_->open_index;
    for (size_t i = 0; i < p.d_->x.size(); i++) {
        float x = d_->x[i], y = d_->y[i], w = d_->w[i];
        p.d_->x[i] = fmaf(t[0], x, fmaf(t[1], y, t[2] * w));
        p.d_->y[i] = fmaf(t[3], x, fmaf(t[4], y, t[5] * w));
        p.d_->w[i] = fmaf(t[6], x, fmaf(t[7], y, t[8] * w));
    }
    return p;
}

// ---- batching: many paths -> one flatten launch ------------------------------------------------------
void FlattenBatch::add(const Path& path, uint32_t slot) {
    const FlattenPlan& pl = path.plan();
    const uint32_t qbase = (uint32_t)x0.size(), sbase = (uint32_t)sp0x.size();
    const size_t pbase = point_commands.size();
    for (size_t i = 0; i < pl.point_commands.size(); i++) {
        uint32_t c = pl.point_commands[i];
        bool tagged = (c & 0x7F800000u) == 0x7F800000u;
        point_commands.push_back(tagged ? ((c & 0xFFC00000u) | ((c & 0x3FFFFFu) + sbase)) : c);
        point_indices.push_back(pl.point_indices[i]);
        quad_indices.push_back(tagged ? 0u : pl.quad_indices[i] + qbase);
        // Path::push_segments_to + SegmentBuffer::push_path (path.rs:689-722, segment.rs:180-198):
        // the point that ends a contour carries no line id
        line_slot.push_back(pl.new_contour[i] ? FORMA_NONE : slot);
    }
    if (!pl.point_commands.empty()) line_slot.back() = FORMA_NONE;
    auto app = [](std::vector<float>& d, const std::vector<float>& s) { d.insert(d.end(), s.begin(), s.end()); };
    app(qx, pl.qx); app(qy, pl.qy); app(qw, pl.qw); app(x0, pl.x0); app(dx_recip, pl.dx_recip); app(k0, pl.k0); app(dk, pl.dk);
    app(curvatures_recip, pl.curvatures_recip); app(partial_curv, pl.partial_curv);
    for (uint32_t s : pl.partial_spline) partial_spline.push_back(s + sbase);
    app(sp0x, pl.sp0x); app(sp0y, pl.sp0y); app(sp2x, pl.sp2x); app(sp2y, pl.sp2y);
    if (path.has_affine()) {
        Affine a; a.first = pbase; a.count = pl.point_commands.size(); std::memcpy(a.m, path.affine(), sizeof a.m);
        affines.push_back(a);
    }
}

void FlattenBatch::tables(forma_flatten_tables_t* t) const {
    t->point_commands = point_commands.data(); t->point_indices = point_indices.data(); t->quad_indices = quad_indices.data();
    t->n_points = point_commands.size();
    t->qx = qx.data(); t->qy = qy.data(); t->qw = qw.data(); t->x0 = x0.data(); t->dx_recip = dx_recip.data();
    t->k0 = k0.data(); t->dk = dk.data(); t->curvatures_recip = curvatures_recip.data();
    t->partial_spline = partial_spline.data(); t->partial_curv = partial_curv.data(); t->n_quads = x0.size();
    t->sp0x = sp0x.data(); t->sp0y = sp0y.data(); t->sp2x = sp2x.data(); t->sp2y = sp2y.data(); t->n_splines = sp0x.size();
}

void FlattenBatch::apply_affines(float* x, float* y) const {    // per-point GeomPresTransform, path.rs:689-706
    for (const Affine& a : affines)
        for (size_t i = a.first; i < a.first + a.count; i++) {
            float px = x[i], py = y[i];
            x[i] = fmaf(a.m[0], px, fmaf(a.m[2], py, a.m[4]));
            y[i] = fmaf(a.m[1], px, fmaf(a.m[3], py, a.m[5]));
        }
}

}  // namespace forma

// ---- flat C wrappers for the ctypes binding ------------------------------------------------------------
using namespace forma;
extern "C" {
void* forma_host_builder_new(void) { return new PathBuilder(); }
void  forma_host_builder_free(void* b) { delete (PathBuilder*)b; }
void  forma_host_move_to(void* b, float x, float y) { ((PathBuilder*)b)->move_to({x, y}); }
void  forma_host_line_to(void* b, float x, float y) { ((PathBuilder*)b)->line_to({x, y}); }
void  forma_host_quad_to(void* b, float ax, float ay, float bx, float by) { ((PathBuilder*)b)->quad_to({ax, ay}, {bx, by}); }
void  forma_host_cubic_to(void* b, float ax, float ay, float bx, float by, float cx, float cy) { ((PathBuilder*)b)->cubic_to({ax, ay}, {bx, by}, {cx, cy}); }
void  forma_host_rat_quad_to(void* b, float ax, float ay, float bx, float by, float w) { ((PathBuilder*)b)->rat_quad_to({ax, ay}, {bx, by}, w); }
void  forma_host_rat_cubic_to(void* b, float ax, float ay, float bx, float by, float cx, float cy, float w1, float w2) {
    ((PathBuilder*)b)->rat_cubic_to({ax, ay}, {bx, by}, {cx, cy}, w1, w2);
}
void* forma_host_build(void* b) { return new Path(((PathBuilder*)b)->build()); }
void  forma_host_path_free(void* p) { delete (Path*)p; }
void* forma_host_path_transform(void* p, const float* t9) { return new Path(((Path*)p)->transform(t9)); }
size_t forma_host_path_points(void* p) { return ((Path*)p)->plan().point_commands.size(); }
// lines the path adds to the geometry store = ids that are Some (SegmentBuffer::len, segment.rs:159-178): every point but
// the last one and the points that end a contour
size_t forma_host_path_lines(void* p) {
    const FlattenPlan& pl = ((Path*)p)->plan();
    size_t n = 0;
    for (size_t i = 0; i + 1 < pl.new_contour.size(); i++) n += pl.new_contour[i] ? 0 : 1;
    return n;
}

void* forma_host_batch_new(void) { return new FlattenBatch(); }
void  forma_host_batch_free(void* b) { delete (FlattenBatch*)b; }
void  forma_host_batch_add(void* b, void* path, uint32_t slot) { ((FlattenBatch*)b)->add(*(Path*)path, slot); }
size_t forma_host_batch_points(void* b) { return ((FlattenBatch*)b)->point_commands.size(); }
// run stage 1 for the whole batch on the GPU: out_x/out_y/out_line_slot have batch_points entries
int forma_host_batch_flatten(void* b_, forma_hip_ctx* ctx, float* out_x, float* out_y, uint32_t* out_line_slot) {
    FlattenBatch* b = (FlattenBatch*)b_;
    forma_flatten_tables_t t;
    b->tables(&t);
    int rc = forma_hip_flatten(ctx, &t, out_x, out_y);
    if (rc) return rc;
    b->apply_affines(out_x, out_y);
    if (!b->line_slot.empty()) std::memcpy(out_line_slot, b->line_slot.data(), b->line_slot.size() * 4);
    return 0;
}
// expose the work items (tests compare them with the oracle's flattener output through the kernel)
void forma_host_batch_tables(void* b, forma_flatten_tables_t* t) { ((FlattenBatch*)b)->tables(t); }
}
