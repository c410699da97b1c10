// lod_builder.cpp -- cluster-LOD DAG builder of libbrmi_scene.so.  Host only.
//
// What it builds is what the reference obtains from meshoptimizer's clusterlod.h driven by
// BR/src/Mesh/ClusterLODUtilities.cpp:5426-5458: a DAG of <= 128-vertex / <= 128-triangle clusters in which every group of
// clusters is simplified to about half its triangles with the group's outer boundary locked, re-clustered, and the new clusters
// remember the group they replace (`refined`).  The rendering rules are clusterlod.h's: draw a cluster when its own group's
// simplification is too coarse (group error over the threshold) and the group it refines is fine enough (error at or under it).
// Errors are merged as max(1.5 x previous, current) (simplify_error_merge_previous = 1.5), a group that cannot drop below 85 %
// of its triangles is terminal (error FLT_MAX), groups aim at 384 clusters and never mix more than 8 refined groups
// (partition_size / partition_max_refined_groups of ClusterLODUtilities.cpp:5450-5456).
//
// The algorithms are this repository's own, chosen for determinism and O(n log n) cost:
//   * clusters and groups come from a balanced k-d split (median cuts of triangle / cluster centres along the longest axis) instead
//     of meshoptimizer's greedy meshlet builder and graph partitioner: every leaf holds n / ceil(n / 128) triangles, i.e. 64 < T <= 128
//     whenever the input has more than 128, and is cut again while it references more than 128 vertices;
//   * the simplifier is a half-edge-collapse QEM on position-welded vertices with per-corner attribute indices ("wedges"): a vertex
//     may only slide along an edge that carries every one of its wedges, so UV / normal seams stay closed and crease corners stay
//     put without a vertex classification pass; open borders collapse along the border only and carry border-plane quadrics; the
//     link condition and a normal-flip test keep the surface manifold.  The reported error is the square root of the largest
//     area-weighted mean squared plane distance any executed collapse had, in mesh units.  (Geometry-only quadrics: the reference
//     also weighs the normals in, clusterlod.h attribute_weights; a documented difference.)
#include "brmi_scene.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <vector>

namespace {

struct Sph { float c[3]; float r; };

inline void sub3(const float* a, const float* b, double* o) { o[0] = (double)a[0] - b[0]; o[1] = (double)a[1] - b[1]; o[2] = (double)a[2] - b[2]; }

// ---- position weld: remap[v] = lowest vertex index with the same position (bitwise, -0 == +0) --------------------------------------
std::vector<uint32_t> weldPositions(const float* pos, size_t n) {
    std::vector<uint32_t> remap(n);
    size_t cap = 16; while (cap < n * 2) cap <<= 1;
    std::vector<uint32_t> table(cap, 0xFFFFFFFFu);
    auto key = [&](size_t v, uint32_t k[3]) { for (int q = 0; q < 3; q++) { float f = pos[v * 3 + q]; if (f == 0.0f) f = 0.0f; std::memcpy(&k[q], &f, 4); } };
    for (size_t v = 0; v < n; v++) {
        uint32_t k[3]; key(v, k);
        uint64_t h = (uint64_t)k[0] * 0x9E3779B185EBCA87ull ^ (uint64_t)k[1] * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)k[2] * 0x165667B19E3779F9ull;
        h ^= h >> 29;
        size_t slot = (size_t)h & (cap - 1);
        for (;;) {
            const uint32_t o = table[slot];
            if (o == 0xFFFFFFFFu) { table[slot] = (uint32_t)v; remap[v] = (uint32_t)v; break; }
            uint32_t ko[3]; key(o, ko);
            if (ko[0] == k[0] && ko[1] == k[1] && ko[2] == k[2]) { remap[v] = o; break; }
            slot = (slot + 1) & (cap - 1);
        }
    }
    return remap;
}

// ---- bounding spheres (Ritter: extreme pair along the axes, then grow) ---------------------------------------------------------------
Sph sphereOfSpheres(const Sph* s, size_t n) {
    if (n == 0) return Sph{{0, 0, 0}, 0};
    size_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (size_t i = 1; i < n; i++) for (int a = 0; a < 3; a++) {
        if (s[i].c[a] - s[i].r < s[lo[a]].c[a] - s[lo[a]].r) lo[a] = i;
        if (s[i].c[a] + s[i].r > s[hi[a]].c[a] + s[hi[a]].r) hi[a] = i;
    }
    int best = 0; double bestD = -1;
    for (int a = 0; a < 3; a++) {
        double d[3]; sub3(s[hi[a]].c, s[lo[a]].c, d);
        const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + s[hi[a]].r + s[lo[a]].r;
        if (len > bestD) { bestD = len; best = a; }
    }
    const Sph &A = s[lo[best]], &B = s[hi[best]];
    double d[3]; sub3(B.c, A.c, d);
    const double dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    double c[3], r;
    if (dist + B.r <= A.r) { c[0] = A.c[0]; c[1] = A.c[1]; c[2] = A.c[2]; r = A.r; }
    else if (dist + A.r <= B.r) { c[0] = B.c[0]; c[1] = B.c[1]; c[2] = B.c[2]; r = B.r; }
    else { r = 0.5 * (dist + A.r + B.r); const double t = dist > 0 ? (r - A.r) / dist : 0.0; for (int q = 0; q < 3; q++) c[q] = A.c[q] + d[q] * t; }
    for (size_t i = 0; i < n; i++) {
        const double e[3] = {s[i].c[0] - c[0], s[i].c[1] - c[1], s[i].c[2] - c[2]};
        const double di = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        if (di + s[i].r > r) {
            const double nr = 0.5 * (r + di + s[i].r);
            if (di > 0) { const double t = (nr - r) / di; for (int q = 0; q < 3; q++) c[q] += e[q] * t; }
            r = nr;
        }
    }
    Sph out; for (int q = 0; q < 3; q++) out.c[q] = (float)c[q];
    // float rounding of the centre must not leave a member outside
    double rr = 0;
    for (size_t i = 0; i < n; i++) { double e[3]; sub3(s[i].c, out.c, e); rr = std::max(rr, std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + (double)s[i].r); }
    out.r = std::nextafter((float)(rr * (1.0 + 1e-6)), FLT_MAX);
    return out;
}

Sph sphereOfVertices(const float* pos, const uint32_t* idx, size_t n) {
    std::vector<Sph> pts(n);
    for (size_t i = 0; i < n; i++) { pts[i].c[0] = pos[(size_t)idx[i] * 3]; pts[i].c[1] = pos[(size_t)idx[i] * 3 + 1]; pts[i].c[2] = pos[(size_t)idx[i] * 3 + 2]; pts[i].r = 0; }
    return sphereOfSpheres(pts.data(), n);
}

// ---- balanced k-d split: items [begin, end) of `order` into leaves of at most `leafMax` items; `accept` may ask for a further cut ----
template <typename Accept, typename Emit>
void kdSplit(std::vector<uint32_t>& order, size_t begin, size_t end, const float* centres, size_t leafMax, const Accept& accept, const Emit& emit) {
    const size_t n = end - begin;
    if (n == 0) return;
    if (n == 1 || (n <= leafMax && accept(order.data() + begin, n))) { std::sort(order.begin() + begin, order.begin() + end); emit(order.data() + begin, n); return; }
    const size_t k = std::max<size_t>(2, (n + leafMax - 1) / leafMax), kl = k / 2;
    const size_t nl = std::min(n - 1, std::max<size_t>(1, (n * kl + k / 2) / k));
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = begin; i < end; i++) for (int a = 0; a < 3; a++) { const float v = centres[(size_t)order[i] * 3 + a]; lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v); }
    int axis = 0; for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
    std::nth_element(order.begin() + begin, order.begin() + begin + nl, order.begin() + end, [&](uint32_t x, uint32_t y) {
        const float cx = centres[(size_t)x * 3 + axis], cy = centres[(size_t)y * 3 + axis];
        return cx < cy || (cx == cy && x < y);
    });
    kdSplit(order, begin, begin + nl, centres, leafMax, accept, emit);
    kdSplit(order, begin + nl, end, centres, leafMax, accept, emit);
}

struct Cluster {
    std::vector<uint32_t> indices;     // original vertex ids, 3 per triangle
    uint32_t vertexCount = 0;
    int refined = -1;
    Sph lodBounds{{0, 0, 0}, 0};       // sphere the LOD decision of this cluster's *source* group uses (nested up the DAG)
    float error = 0.0f;                // error of the group whose simplification produced this cluster (0 for the input mesh)
};

constexpr size_t kMaxVerts = 128, kMaxTris = 128, kGroupTarget = 384, kMaxRefinedPerGroup = 8;
constexpr double kSimplifyRatio = 0.5, kSimplifyThreshold = 0.85, kErrorMergePrevious = 1.5;

// triangles -> clusters of <= 128 triangles / <= 128 vertices
void clusterize(const float* pos, const uint32_t* idx, size_t indexCount, std::vector<uint32_t>& stamp, uint32_t& epoch, std::vector<Cluster>& out) {
    const size_t T = indexCount / 3;
    if (T == 0) return;
    std::vector<float> centres(T * 3);
    for (size_t t = 0; t < T; t++) for (int a = 0; a < 3; a++)
        centres[t * 3 + a] = (pos[(size_t)idx[t * 3] * 3 + a] + pos[(size_t)idx[t * 3 + 1] * 3 + a] + pos[(size_t)idx[t * 3 + 2] * 3 + a]) * (1.0f / 3.0f);
    std::vector<uint32_t> order(T);
    for (size_t t = 0; t < T; t++) order[t] = (uint32_t)t;
    auto uniqueVerts = [&](const uint32_t* tris, size_t n) {
        epoch++; size_t u = 0;
        for (size_t i = 0; i < n; i++) for (int c = 0; c < 3; c++) { const uint32_t v = idx[(size_t)tris[i] * 3 + c]; if (stamp[v] != epoch) { stamp[v] = epoch; u++; } }
        return u;
    };
    kdSplit(order, 0, T, centres.data(), kMaxTris,
        [&](const uint32_t* tris, size_t n) { return uniqueVerts(tris, n) <= kMaxVerts; },
        [&](const uint32_t* tris, size_t n) {
            Cluster c; c.indices.reserve(n * 3);
            for (size_t i = 0; i < n; i++) for (int q = 0; q < 3; q++) c.indices.push_back(idx[(size_t)tris[i] * 3 + q]);
            c.vertexCount = (uint32_t)uniqueVerts(tris, n);
            out.push_back(std::move(c));
        });
}

// ---- simplifier -----------------------------------------------------------------------------------------------------------------
struct Quadric {
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0, b0 = 0, b1 = 0, b2 = 0, c = 0, w = 0;
    void addPlane(const double n[3], double d, double weight) {
        a00 += weight * n[0] * n[0]; a01 += weight * n[0] * n[1]; a02 += weight * n[0] * n[2]; a11 += weight * n[1] * n[1]; a12 += weight * n[1] * n[2]; a22 += weight * n[2] * n[2];
        b0 += weight * n[0] * d; b1 += weight * n[1] * d; b2 += weight * n[2] * d; c += weight * d * d; w += weight;
    }
    void add(const Quadric& o) { a00 += o.a00; a01 += o.a01; a02 += o.a02; a11 += o.a11; a12 += o.a12; a22 += o.a22; b0 += o.b0; b1 += o.b1; b2 += o.b2; c += o.c; w += o.w; }
    double eval(const double p[3]) const {
        return a00 * p[0] * p[0] + a11 * p[1] * p[1] + a22 * p[2] * p[2] + 2.0 * (a01 * p[0] * p[1] + a02 * p[0] * p[2] + a12 * p[1] * p[2]) + 2.0 * (b0 * p[0] + b1 * p[1] + b2 * p[2]) + c;
    }
};

struct Simplifier {
    struct Tri { uint32_t v[3]; uint32_t o[3]; bool alive; };
    const float* pos;
    std::vector<Tri> tris;
    std::vector<std::vector<uint32_t>> vtris;      // local vertex -> incident triangles (alive ones are filtered on use)
    std::vector<uint32_t> localToWelded;           // local vertex -> welded (global) vertex id
    std::vector<Quadric> Q;
    std::vector<uint8_t> locked, border;
    std::vector<uint32_t> stamp;
    size_t aliveTris = 0;
    double maxErr2 = 0;

    const float* P(uint32_t lv) const { return pos + (size_t)localToWelded[lv] * 3; }

    struct Cand { double cost; uint32_t a, b, sa, sb; };
    struct Worse { bool operator()(const Cand& x, const Cand& y) const { return x.cost > y.cost || (x.cost == y.cost && (x.a > y.a || (x.a == y.a && x.b > y.b))); } };
    std::priority_queue<Cand, std::vector<Cand>, Worse> heap;

    static bool triNormal(const float* a, const float* b, const float* c, double n[3], double& area2) {
        double e1[3], e2[3]; sub3(b, a, e1); sub3(c, a, e2);
        n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
        area2 = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (area2 <= 0) return false;
        n[0] /= area2; n[1] /= area2; n[2] /= area2;
        return true;
    }

    void neighbours(uint32_t v, std::vector<uint32_t>& out) const {
        out.clear();
        for (uint32_t t : vtris[v]) if (tris[t].alive) for (int c = 0; c < 3; c++) if (tris[t].v[c] != v) out.push_back(tris[t].v[c]);
        std::sort(out.begin(), out.end()); out.erase(std::unique(out.begin(), out.end()), out.end());
    }
    bool isBorderEdge(uint32_t a, uint32_t b) const {
        int n = 0;
        for (uint32_t t : vtris[a]) if (tris[t].alive) { const Tri& tr = tris[t]; if (tr.v[0] == b || tr.v[1] == b || tr.v[2] == b) n++; }
        return n == 1;
    }

    double cost(uint32_t a, uint32_t b) const {
        const float* pb = P(b); const double p[3] = {pb[0], pb[1], pb[2]};
        Quadric q = Q[a]; q.add(Q[b]);
        const double e = std::max(0.0, q.eval(p)) / std::max(q.w, 1e-300);
        double d[3]; sub3(P(a), pb, d);
        return e + 1e-10 * (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);       // ties (flat regions): shortest edge first
    }
    void pushEdge(uint32_t a, uint32_t b) {
        if (!locked[a]) heap.push({cost(a, b), a, b, stamp[a], stamp[b]});
        if (!locked[b]) heap.push({cost(b, a), b, a, stamp[b], stamp[a]});
    }

    void init(const float* positions, const std::vector<uint32_t>& indices, const std::vector<uint32_t>& remap, const std::vector<uint8_t>& weldedLocked,
              std::vector<uint32_t>& scratchLocal /* welded id -> local id, 0xFFFFFFFF-filled */) {
        pos = positions;
        const size_t T = indices.size() / 3;
        tris.resize(T);
        for (size_t t = 0; t < T; t++) {
            for (int c = 0; c < 3; c++) {
                const uint32_t o = indices[t * 3 + c], w = remap[o];
                uint32_t l = scratchLocal[w];
                if (l == 0xFFFFFFFFu) { l = (uint32_t)localToWelded.size(); scratchLocal[w] = l; localToWelded.push_back(w); }
                tris[t].v[c] = l; tris[t].o[c] = o;
            }
            tris[t].alive = !(tris[t].v[0] == tris[t].v[1] || tris[t].v[1] == tris[t].v[2] || tris[t].v[0] == tris[t].v[2]);
        }
        for (uint32_t w : localToWelded) scratchLocal[w] = 0xFFFFFFFFu;      // leave the scratch table clean for the next group
        const size_t V = localToWelded.size();
        vtris.assign(V, {}); Q.assign(V, Quadric()); locked.assign(V, 0); border.assign(V, 0); stamp.assign(V, 0);
        for (size_t v = 0; v < V; v++) locked[v] = weldedLocked[localToWelded[v]];
        for (size_t t = 0; t < T; t++) if (tris[t].alive) { aliveTris++; for (int c = 0; c < 3; c++) vtris[tris[t].v[c]].push_back((uint32_t)t); }
        // plane quadrics, area weighted
        for (size_t t = 0; t < T; t++) if (tris[t].alive) {
            const Tri& tr = tris[t]; double n[3], a2;
            if (!triNormal(P(tr.v[0]), P(tr.v[1]), P(tr.v[2]), n, a2)) continue;
            const float* p0 = P(tr.v[0]); const double d = -(n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2]);
            for (int c = 0; c < 3; c++) Q[tr.v[c]].addPlane(n, d, 0.5 * a2);
        }
        // open borders: a plane through the edge, perpendicular to its triangle, keeps the outline in place
        std::vector<uint32_t> nb;
        for (uint32_t a = 0; a < V; a++) {
            neighbours(a, nb);
            for (uint32_t b : nb) if (a < b && isBorderEdge(a, b)) {
                border[a] = border[b] = 1;
                for (uint32_t t : vtris[a]) if (tris[t].alive) {
                    const Tri& tr = tris[t];
                    if (tr.v[0] != b && tr.v[1] != b && tr.v[2] != b) continue;
                    double n[3], a2; if (!triNormal(P(tr.v[0]), P(tr.v[1]), P(tr.v[2]), n, a2)) continue;
                    double e[3]; sub3(P(b), P(a), e);
                    double m[3] = {e[1] * n[2] - e[2] * n[1], e[2] * n[0] - e[0] * n[2], e[0] * n[1] - e[1] * n[0]};
                    const double len = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]); if (len <= 0) continue;
                    m[0] /= len; m[1] /= len; m[2] /= len;
                    const float* pa = P(a); const double d = -(m[0] * pa[0] + m[1] * pa[1] + m[2] * pa[2]);
                    const double wgt = 2.0 * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);   // an edge-length^2 strip on either side
                    Q[a].addPlane(m, d, wgt); Q[b].addPlane(m, d, wgt);
                }
            }
        }
        for (uint32_t a = 0; a < V; a++) { neighbours(a, nb); for (uint32_t b : nb) if (a < b) pushEdge(a, b); }
    }

    // collapse a -> b (b keeps its position); returns false when the move is not allowed
    bool tryCollapse(uint32_t a, uint32_t b, std::vector<uint32_t>& nbA, std::vector<uint32_t>& nbB) {
        if (locked[a]) return false;
        // wedges: every attribute index used at `a` must also be used by a triangle on the edge, which names its replacement at `b`
        uint32_t mapFrom[2], mapTo[2]; int nMap = 0, edgeTris = 0;
        for (uint32_t t : vtris[a]) if (tris[t].alive) {
            const Tri& tr = tris[t]; int ca = -1, cb = -1;
            for (int c = 0; c < 3; c++) { if (tr.v[c] == a) ca = c; else if (tr.v[c] == b) cb = c; }
            if (cb < 0) continue;
            edgeTris++;
            bool known = false;
            for (int k = 0; k < nMap; k++) if (mapFrom[k] == tr.o[ca]) { if (mapTo[k] != tr.o[cb]) return false; known = true; }
            if (!known) { if (nMap == 2) return false; mapFrom[nMap] = tr.o[ca]; mapTo[nMap] = tr.o[cb]; nMap++; }
        }
        if (edgeTris == 0 || edgeTris > 2) return false;
        if (border[a] && !(edgeTris == 1 && border[b])) return false;        // a border vertex slides along the border only
        if (!border[a] && edgeTris != 2) return false;
        for (uint32_t t : vtris[a]) if (tris[t].alive) {
            const Tri& tr = tris[t]; int ca = tr.v[0] == a ? 0 : (tr.v[1] == a ? 1 : 2);
            bool covered = false; for (int k = 0; k < nMap; k++) if (mapFrom[k] == tr.o[ca]) covered = true;
            if (!covered) return false;
        }
        // link condition: the two rings may only meet in the vertices opposite the edge
        neighbours(a, nbA); neighbours(b, nbB);
        size_t common = 0; { size_t i = 0, j = 0; while (i < nbA.size() && j < nbB.size()) { if (nbA[i] == nbB[j]) { common++; i++; j++; } else if (nbA[i] < nbB[j]) i++; else j++; } }
        if (common != (size_t)edgeTris) return false;
        // no triangle may flip or degenerate
        const float* pb = P(b);
        for (uint32_t t : vtris[a]) if (tris[t].alive) {
            const Tri& tr = tris[t];
            if (tr.v[0] == b || tr.v[1] == b || tr.v[2] == b) continue;
            const float* p[3] = {P(tr.v[0]), P(tr.v[1]), P(tr.v[2])};
            double n0[3], n1[3], a0, a1;
            if (!triNormal(p[0], p[1], p[2], n0, a0)) continue;
            for (int c = 0; c < 3; c++) if (tr.v[c] == a) p[c] = pb;
            if (!triNormal(p[0], p[1], p[2], n1, a1)) return false;
            if (n0[0] * n1[0] + n0[1] * n1[1] + n0[2] * n1[2] < 0.2 || a1 < 1e-12 * a0) return false;
        }
        // execute
        for (uint32_t t : vtris[a]) if (tris[t].alive) {
            Tri& tr = tris[t];
            if (tr.v[0] == b || tr.v[1] == b || tr.v[2] == b) { tr.alive = false; aliveTris--; continue; }
            for (int c = 0; c < 3; c++) if (tr.v[c] == a) { tr.v[c] = b; for (int k = 0; k < nMap; k++) if (mapFrom[k] == tr.o[c]) { tr.o[c] = mapTo[k]; break; } }
            vtris[b].push_back(t);
        }
        vtris[a].clear();
        {   // drop dead triangles from b's list now and then so that the lists stay short
            auto& l = vtris[b]; size_t k = 0; for (uint32_t t : l) if (tris[t].alive) l[k++] = t; l.resize(k);
        }
        Q[b].add(Q[a]);
        stamp[a]++; stamp[b]++;
        neighbours(b, nbB);
        for (uint32_t x : nbB) pushEdge(b, x);
        return true;
    }

    // returns the simplified index list (original vertex ids) and the error
    void run(size_t targetTris, std::vector<uint32_t>& outIndices, float& outError) {
        std::vector<uint32_t> nbA, nbB;
        while (aliveTris > targetTris && !heap.empty()) {
            const Cand c = heap.top(); heap.pop();
            if (stamp[c.a] != c.sa || stamp[c.b] != c.sb) continue;
            const float* pb = P(c.b); const double p[3] = {pb[0], pb[1], pb[2]};
            Quadric q = Q[c.a]; q.add(Q[c.b]);
            const double e2 = std::max(0.0, q.eval(p)) / std::max(q.w, 1e-300);
            if (tryCollapse(c.a, c.b, nbA, nbB)) maxErr2 = std::max(maxErr2, e2);
        }
        outIndices.clear();
        for (const Tri& t : tris) if (t.alive) { outIndices.push_back(t.o[0]); outIndices.push_back(t.o[1]); outIndices.push_back(t.o[2]); }
        outError = (float)std::sqrt(maxErr2);
    }
};

struct Dag {
    std::vector<brmi_dag_group> groups; std::vector<brmi_dag_cluster> clusters; std::vector<uint32_t> vertexRefs; std::vector<uint8_t> triangles;
};

// meshlet-local indices of one cluster: vertexRefs in first-use order
void emitCluster(Dag& dag, const float* pos, const Cluster& c, int groupId, std::vector<uint32_t>& stamp, std::vector<uint32_t>& localOf, uint32_t& epoch) {
    brmi_dag_cluster o{};
    o.group = groupId; o.refined = c.refined;
    o.firstVertex = (uint32_t)dag.vertexRefs.size(); o.firstTriangleByte = (uint32_t)dag.triangles.size(); o.triangleCount = (uint32_t)(c.indices.size() / 3);
    epoch++;
    uint32_t unique = 0;
    for (uint32_t v : c.indices) {
        if (stamp[v] != epoch) { stamp[v] = epoch; localOf[v] = unique++; dag.vertexRefs.push_back(v); }
        dag.triangles.push_back((uint8_t)localOf[v]);
    }
    o.vertexCount = unique;
    const Sph own = sphereOfVertices(pos, dag.vertexRefs.data() + o.firstVertex, unique);     // culling sphere: the cluster's own geometry
    std::memcpy(o.center, own.c, 12); o.radius = own.r; o.error = c.error;
    dag.clusters.push_back(o);
}

int emitGroup(Dag& dag, const float* pos, const std::vector<Cluster>& clusters, const std::vector<uint32_t>& members, int depth, const Sph& bounds, float error,
              std::vector<uint32_t>& stamp, std::vector<uint32_t>& localOf, uint32_t& epoch) {
    brmi_dag_group g{};
    g.depth = depth; std::memcpy(g.center, bounds.c, 12); g.radius = bounds.r; g.error = error;
    g.firstCluster = (uint32_t)dag.clusters.size(); g.clusterCount = (uint32_t)members.size();
    const int id = (int)dag.groups.size();
    for (uint32_t m : members) emitCluster(dag, pos, clusters[m], id, stamp, localOf, epoch);
    dag.groups.push_back(g);
    return id;
}

}  // namespace

extern "C" {

int brmi_lod_build(void* /*user*/, const float* positions, size_t vertexCount, const uint32_t* indices, size_t indexCount, const float* /*normals*/, brmi_dag* out) {
    if (!out) return -1;
    std::memset(out, 0, sizeof(*out));
    if (!positions || !indices || vertexCount == 0 || indexCount < 3 || indexCount % 3) return -1;
    for (size_t i = 0; i < indexCount; i++) if (indices[i] >= vertexCount) return -1;
    Dag* dag = new Dag();
    const std::vector<uint32_t> remap = weldPositions(positions, vertexCount);
    std::vector<uint32_t> stamp(vertexCount, 0), localOf(vertexCount, 0), scratchLocal(vertexCount, 0xFFFFFFFFu);
    uint32_t epoch = 0;
    std::vector<Cluster> clusters;
    clusterize(positions, indices, indexCount, stamp, epoch, clusters);
    for (Cluster& c : clusters) { c.lodBounds = sphereOfVertices(positions, c.indices.data(), c.indices.size()); c.error = 0.0f; }
    std::vector<uint32_t> pending(clusters.size());
    for (size_t i = 0; i < clusters.size(); i++) pending[i] = (uint32_t)i;
    std::vector<uint8_t> weldedLocked(vertexCount, 0);
    std::vector<int32_t> firstGroupOf(vertexCount, -1);
    int depth = 0;
    while (pending.size() > 1) {
        // groups: k-d split of the cluster centres, ~384 clusters each, at most 8 distinct refined groups per group
        std::vector<float> centres(pending.size() * 3);
        for (size_t i = 0; i < pending.size(); i++) std::memcpy(&centres[i * 3], clusters[pending[i]].lodBounds.c, 12);
        {   // centres of the cluster geometry, not of the (shared) LOD sphere
            for (size_t i = 0; i < pending.size(); i++) {
                const Cluster& c = clusters[pending[i]]; double s[3] = {0, 0, 0};
                for (uint32_t v : c.indices) { s[0] += positions[(size_t)v * 3]; s[1] += positions[(size_t)v * 3 + 1]; s[2] += positions[(size_t)v * 3 + 2]; }
                for (int a = 0; a < 3; a++) centres[i * 3 + a] = (float)(s[a] / (double)c.indices.size());
            }
        }
        std::vector<uint32_t> order(pending.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
        std::vector<std::vector<uint32_t>> groups;
        kdSplit(order, 0, order.size(), centres.data(), kGroupTarget + kGroupTarget / 3,
            [&](const uint32_t* items, size_t n) {
                int keys[kMaxRefinedPerGroup + 1]; size_t nk = 0;
                for (size_t i = 0; i < n; i++) {
                    const int r = clusters[pending[items[i]]].refined; bool seen = false;
                    for (size_t k = 0; k < nk; k++) if (keys[k] == r) { seen = true; break; }
                    if (!seen) { if (nk == kMaxRefinedPerGroup) return false; keys[nk++] = r; }
                }
                return true;
            },
            [&](const uint32_t* items, size_t n) { std::vector<uint32_t> g(n); for (size_t i = 0; i < n; i++) g[i] = pending[items[i]]; groups.push_back(std::move(g)); });
        // (kdSplit aims at k = ceil(n / 512) leaves; re-balance towards the 384 target by asking for leaves of at most 512)
        pending.clear();
        // lock every position that more than one group touches: the simplified groups must keep fitting together
        std::fill(weldedLocked.begin(), weldedLocked.end(), 0); std::fill(firstGroupOf.begin(), firstGroupOf.end(), -1);
        for (size_t gi = 0; gi < groups.size(); gi++) for (uint32_t ci : groups[gi]) for (uint32_t v : clusters[ci].indices) {
            const uint32_t w = remap[v];
            if (firstGroupOf[w] < 0) firstGroupOf[w] = (int32_t)gi; else if (firstGroupOf[w] != (int32_t)gi) weldedLocked[w] = 1;
        }
        for (size_t gi = 0; gi < groups.size(); gi++) {
            const std::vector<uint32_t>& group = groups[gi];
            std::vector<uint32_t> merged;
            for (uint32_t ci : group) merged.insert(merged.end(), clusters[ci].indices.begin(), clusters[ci].indices.end());
            std::vector<Sph> parts; float prevError = 0.0f;
            for (uint32_t ci : group) { parts.push_back(clusters[ci].lodBounds); prevError = std::max(prevError, clusters[ci].error); }
            const Sph bounds = sphereOfSpheres(parts.data(), parts.size());
            const size_t targetTris = std::max<size_t>(1, (size_t)((double)(merged.size() / 3) * kSimplifyRatio));
            std::vector<uint32_t> simplified; float err = 0.0f;
            {
                Simplifier s;
                s.init(positions, merged, remap, weldedLocked, scratchLocal);
                s.run(targetTris, simplified, err);
            }
            if (simplified.size() < 3 || (double)simplified.size() > (double)merged.size() * kSimplifyThreshold) {
                emitGroup(*dag, positions, clusters, group, depth, bounds, FLT_MAX, stamp, localOf, epoch);       // stuck: terminal
                continue;
            }
            const float groupError = std::max(prevError * (float)kErrorMergePrevious, err);
            const int id = emitGroup(*dag, positions, clusters, group, depth, bounds, groupError, stamp, localOf, epoch);
            const size_t first = clusters.size();
            clusterize(positions, simplified.data(), simplified.size(), stamp, epoch, clusters);
            for (uint32_t ci : group) std::vector<uint32_t>().swap(clusters[ci].indices);
            for (size_t k = first; k < clusters.size(); k++) { clusters[k].refined = id; clusters[k].lodBounds = bounds; clusters[k].error = groupError; pending.push_back((uint32_t)k); }
        }
        depth++;
    }
    if (!pending.empty()) {
        const Cluster& c = clusters[pending[0]];
        emitGroup(*dag, positions, clusters, pending, depth, c.lodBounds, FLT_MAX, stamp, localOf, epoch);
    }
    out->groups = dag->groups.data(); out->groupCount = (uint32_t)dag->groups.size();
    out->clusters = dag->clusters.data(); out->clusterCount = (uint32_t)dag->clusters.size();
    out->vertexRefs = dag->vertexRefs.data(); out->vertexRefCount = (uint32_t)dag->vertexRefs.size();
    out->triangles = dag->triangles.data(); out->triangleBytes = (uint32_t)dag->triangles.size();
    out->owner = dag;
    return 0;
}

void brmi_lod_release(void* /*user*/, brmi_dag* dag) {
    if (!dag) return;
    delete static_cast<Dag*>(dag->owner);
    std::memset(dag, 0, sizeof(*dag));
}

}  // extern "C"
