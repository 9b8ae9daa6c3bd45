/*
 * bvh.h -- host-side acceleration-structure build for path_hip.
 *
 * The reference answers ray queries with a SAH kd-tree (include/mitsuba/render/gkdtree.h,
 * sahkdtree3.h); a kd-tree's pointer-chasing, duplicated references and 48-entry Havran stack
 * are a poor fit for 64-wide wavefronts.  The MI355X design instead uses a binary BVH built
 * with binned SAH on the host, flattened into 64-byte nodes that hold BOTH children's boxes
 * (one node visit = four coalescable 16-byte loads, two slab tests), with triangles stored in
 * leaf order as 48-byte Wald records (the reference's TriAccel arithmetic,
 * include/mitsuba/render/triaccel.h:61-93, so that (t,u,v) are bit-identical to the CPU path).
 * Closest-hit results do not depend on the structure, only on the triangle test.
 *
 * The binary tree is then collapsed into a 4-wide BVH (a child that is an inner node is replaced by
 * its two children, largest surface area first) so that one 128-byte node -- exactly one cache line,
 * eight coalescable 16-byte loads -- yields four slab tests per dependent memory round trip.
 *
 * BVH4 node layout (8 x float4, children in the four lanes of each vector):
 *   n0 = min.x[4]  n1 = min.y[4]  n2 = min.z[4]  n3 = max.x[4]  n4 = max.y[4]  n5 = max.z[4]
 *   n6 = bits(child ref[4])   n7 = unused
 *   empty child slots have min = +inf, max = -inf (never hit).
 * (intermediate BVH2 node, 4 x float4: l.min.xyz l.max.x | l.max.yz r.min.xy | r.min.z r.max.xyz | refs)
 * child reference: >= 0 inner-node index; < 0 leaf: ~ref = (firstTri << 3) | (count - 1), count in 1..8.
 * Triangle record (3 x float4): (bits(k), n_u, n_v, n_d) (a_u, a_v, b_nu, b_nv) (c_nu, c_nv, bits(globalPrim), 0)
 */
#pragma once
#include <vector>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <algorithm>
#include <chrono>
#include <functional>
#include <thread>
#include <exception>

namespace pt {

struct BuildTri { float bmin[3], bmax[3], c[3]; uint32_t prim; };
constexpr uint32_t SPATIAL_MIN_TRIS = 4096;     /* scenes from this size on are built with spatial splits */

struct HostBVH {
    /* 8-wide tree with quantised child boxes for the big-scene ray kernels (k_wide.h), after Ylitie, Karras & Laine,
       "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs" (HPG 2017): 80-byte nodes, see buildWide() */
    std::vector<uint32_t> wnodes; /* 20 dwords per node */
    std::vector<float> wtris;     /* 12 floats per triangle record, grouped per wide node (each node's leaf triangles are consecutive) */
    uint32_t nWNodes = 0, wMaxDepth = 0, nWTris = 0; float wSahCost = 0;
    std::vector<float> nodes;     /* 32 floats per BVH4 node */
    std::vector<float> nodes2;    /* 16 floats per intermediate BVH2 node */
    uint32_t nNodes2 = 0;
    std::vector<float> tris;      /* 12 floats per triangle record, leaf order */
    uint32_t nNodes = 0, nLeaves = 0, nTriRefs = 0, maxDepth = 0;
    float sceneMin[3], sceneMax[3];       /* enlarged like gkdtree.h:1213-1220 */
    float tightMin[3], tightMax[3];
    float sahCost = 0, buildMs = 0;
    int32_t rootRef = 0;
};

namespace detail {

inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* TriAccel::load, triaccel.h:61-93 -- returns false for degenerate triangles (k = 3) */
inline bool waldLoad(const float *A, const float *B, const float *C, uint32_t prim, float *out12) {
    static const int waldModulo[4] = { 1, 2, 0, 1 };
    float b[3] = { C[0] - A[0], C[1] - A[1], C[2] - A[2] };
    float c[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] };
    float N[3] = { c[1] * b[2] - c[2] * b[1], c[2] * b[0] - c[0] * b[2], c[0] * b[1] - c[1] * b[0] };
    uint32_t k = 0;
    for (int j = 0; j < 3; j++)
        if (std::fabs(N[j]) > std::fabs(N[k])) k = j;
    uint32_t u = waldModulo[k], v = waldModulo[k + 1];
    const float n_k = N[k], denom = b[u] * c[v] - b[v] * c[u];
    if (denom == 0) return false;
    out12[0] = bits2f(k);
    out12[1] = N[u] / n_k;
    out12[2] = N[v] / n_k;
    out12[3] = (A[0] * N[0] + A[1] * N[1] + A[2] * N[2]) / n_k;
    out12[4] = A[u];
    out12[5] = A[v];
    out12[6] = b[u] / denom;
    out12[7] = -b[v] / denom;
    out12[8] = c[v] / denom;
    out12[9] = -c[u] / denom;
    out12[10] = bits2f(prim);
    out12[11] = 0.0f;
    return true;
}

struct Box {
    float mn[3], mx[3];
    void reset() { for (int i = 0; i < 3; ++i) { mn[i] = INFINITY; mx[i] = -INFINITY; } }
    void grow(const float *a, const float *b) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], a[i]); mx[i] = std::max(mx[i], b[i]); } }
    void growPt(const float *p) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); } }
    float area() const {
        float d[3] = { mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2] };
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.0f;
        return 2.0f * (d[0] * d[1] + d[1] * d[2] + d[0] * d[2]);
    }
};

struct ChildRef { int32_t ref; Box box; };      /* a child of a BVH2 node: reference (>= 0 inner, < 0 leaf) + padded box */

struct Builder {
    std::vector<BuildTri> &T;
    HostBVH &out;
    const float *positions; const uint32_t *indices;
    static constexpr int NBINS = 32;
    int MAX_LEAF = 4;
    float C_TRAV = 1.0f, C_ISECT = 1.0f;
    double sah = 0;
    /* spatial splits (Stich, Friedrich & Dietrich, "Spatial Splits in Bounding Volume Hierarchies", HPG 2009): scenes with long
       thin triangles (architecture: floors, walls, beams next to finely tessellated detail) */
    static constexpr int SBINS = 32;
    bool spatial = false;
    float alpha = 1e-5f;               /* a spatial split is only tried when the object split's children overlap by more than alpha * root area */
    double rootArea = 1;
    size_t slack = 0;                  /* references spatial splits may still ADD below this node (the whole tree: at most 0.5 per triangle) */
    int dealLevels = 5;                /* buildSpatial: top levels of the tree on which the reference budget is dealt to the two sides in proportion to their
                                          reference counts -- on every host, threaded or not, so that the tree is the same on every machine */
    int parallelLevels = 0;            /* buildSpatial: levels of the tree whose two subtrees are built on two threads (round 3: the builder was
                                          single-threaded and 8 x slower than the plain SAH build on a 250 k-triangle scene) */
    static constexpr size_t PAR_MIN_REFS = 8192;

    Builder(std::vector<BuildTri> &t, HostBVH &o, const float *p, const uint32_t *i) : T(t), out(o), positions(p), indices(i) {
        if (const char *e = getenv("PHIP_BVH_MAXLEAF")) MAX_LEAF = std::min(8, std::max(1, atoi(e)));     /* experiment hooks */
        if (const char *e = getenv("PHIP_BVH_CTRAV")) C_TRAV = (float) atof(e);
        if (const char *e = getenv("PHIP_BVH_ALPHA")) alpha = (float) atof(e);
    }

    /* bounds of (triangle `prim` restricted to lo <= x[axis] <= hi), intersected with `within`; false if nothing is left.
       Computed in double and widened by one float ulp: conservative (and the node boxes are padded on top, pad()). */
    bool clippedBounds(uint32_t prim, int axis, double lo, double hi, const Box &within, Box &outb) const {
        double v[3][3];
        for (int k = 0; k < 3; ++k) { const float *p = positions + 3 * (size_t) indices[3 * (size_t) prim + k]; v[k][0] = p[0]; v[k][1] = p[1]; v[k][2] = p[2]; }
        double mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
        auto add = [&](const double *q) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], q[a]); mx[a] = std::max(mx[a], q[a]); } };
        for (int k = 0; k < 3; ++k) {
            const double *p0 = v[k], *p1 = v[(k + 1) % 3];
            if (p0[axis] >= lo && p0[axis] <= hi) add(p0);
            const double planes[2] = { lo, hi };
            for (int s2 = 0; s2 < 2; ++s2) {
                const double pl = planes[s2];
                if (!std::isfinite(pl)) continue;
                if ((p0[axis] < pl && p1[axis] > pl) || (p0[axis] > pl && p1[axis] < pl)) {
                    const double t = (pl - p0[axis]) / (p1[axis] - p0[axis]);
                    double q[3]; for (int a = 0; a < 3; ++a) q[a] = p0[a] + t * (p1[a] - p0[a]);
                    q[axis] = pl; add(q);
                }
            }
        }
        outb.reset();
        for (int a = 0; a < 3; ++a) {
            if (!(mn[a] <= mx[a])) return false;
            float l = std::nextafterf((float) mn[a], -INFINITY), h = std::nextafterf((float) mx[a], INFINITY);
            l = std::max(l, within.mn[a]); h = std::min(h, within.mx[a]);
            if (!(l <= h)) return false;
            outb.mn[a] = l; outb.mx[a] = h;
        }
        return true;
    }
    static BuildTri makeRef(const Box &b, uint32_t prim) {
        BuildTri r; r.prim = prim;
        for (int a = 0; a < 3; ++a) { r.bmin[a] = b.mn[a]; r.bmax[a] = b.mx[a]; r.c[a] = 0.5f * (b.mn[a] + b.mx[a]); }
        return r;
    }

    /* pads a box so that a hit accepted by the Wald test (which tolerates a few ulp outside the exact triangle) can never be
       culled by the slab test -- including a hit at EXACTLY the current maxt (a tie with a hit found earlier, winsTie in
       dv_scene.h): the slab's entry distance and the Wald distance are rounded differently, a few ulp of t apart, i.e. up to
       ~3e-7 x the distance travelled along the axis.  A pad relative to the coordinate alone vanishes for a flat box in the
       plane x = 0 (round 2: ties on such planes were decided by the traversal order), hence the term in the scene's extent. */
    float extent[3] = { 0, 0, 0 };
    /* ... and, round 6, a term in the CAMERA's position: the slab distance of a plane is (plane - o) * rcp, whose rounding grows with |o|, and the camera's are the only rays that
       start outside the scene box -- a telephoto view from thousands of scene extents away used to be safe only on the BVH4 of the small scenes, whose kernels left the product
       when every scene got the wide tree (tests/test_gpu_parity.py: far camera; the packed leaf table pads its half extents the same way, phip.hip) */
    float camPad[3] = { 0, 0, 0 };
    void pad(Box &b) const {
        /* (the LARGEST extent on every axis: a scene that is flat on one axis -- everything in the plane y = 0 -- must not lose its pad there) */
        const float maxExtent = std::max(extent[0], std::max(extent[1], extent[2]));
        for (int i = 0; i < 3; ++i) {
            float e = 1e-5f * std::max(std::fabs(b.mn[i]), std::fabs(b.mx[i])) + 1e-7f * (b.mx[i] - b.mn[i]) + 2e-6f * maxExtent + camPad[i] + 1e-30f;
            b.mn[i] -= e; b.mx[i] += e;
        }
    }

    int32_t makeLeaf(size_t b, size_t e) { return makeLeaf(T.data() + b, T.data() + e); }
    int32_t makeLeaf(const BuildTri *rb, const BuildTri *re) {
        uint32_t first = (uint32_t) (out.tris.size() / 12);
        uint32_t count = 0;
        for (const BuildTri *it = rb; it != re; ++it) {
            float rec[12];
            const uint32_t p = it->prim;
            const float *A = positions + 3 * (size_t) indices[3 * (size_t) p], *B = positions + 3 * (size_t) indices[3 * (size_t) p + 1],
                        *C = positions + 3 * (size_t) indices[3 * (size_t) p + 2];
            if (!waldLoad(A, B, C, p, rec)) continue;
            out.tris.insert(out.tris.end(), rec, rec + 12);
            ++count;
        }
        out.nLeaves++; out.nTriRefs += count;
        if (count == 0) { /* keep a well-formed (never-hit) leaf: one record with k = 3 */
            float rec[12] = { 0 }; rec[0] = bits2f(3u); rec[10] = bits2f(0xFFFFFFFFu);
            out.tris.insert(out.tris.end(), rec, rec + 12);
            count = 1;
        }
        return ~(int32_t) ((first << 3) | (count - 1));
    }

    /* returns child reference; box = bounds of the subtree */
    int32_t build(size_t b, size_t e, Box &box, uint32_t depth) {
        out.maxDepth = std::max(out.maxDepth, depth);
        box.reset();
        Box cb; cb.reset();
        for (size_t i = b; i < e; ++i) { box.grow(T[i].bmin, T[i].bmax); cb.growPt(T[i].c); }
        const size_t n = e - b;
        if (n == 1) return makeLeaf(b, e);

        /* binned SAH over the centroid bounds */
        float bestCost = INFINITY; int bestAxis = -1, bestBin = -1;
        for (int axis = 0; axis < 3; ++axis) {
            float lo = cb.mn[axis], hi = cb.mx[axis];
            if (!(hi > lo)) continue;
            Box bins[NBINS]; uint32_t cnt[NBINS];
            for (int i = 0; i < NBINS; ++i) { bins[i].reset(); cnt[i] = 0; }
            const float scale = NBINS / (hi - lo);
            for (size_t i = b; i < e; ++i) {
                int k = (int) ((T[i].c[axis] - lo) * scale); k = std::min(std::max(k, 0), NBINS - 1);
                bins[k].grow(T[i].bmin, T[i].bmax); cnt[k]++;
            }
            float rightArea[NBINS]; uint32_t rightCnt[NBINS];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int i = NBINS - 1; i > 0; --i) { acc.grow(bins[i].mn, bins[i].mx); c += cnt[i]; rightArea[i] = acc.area(); rightCnt[i] = c; }
            acc.reset(); c = 0;
            for (int i = 0; i < NBINS - 1; ++i) {
                acc.grow(bins[i].mn, bins[i].mx); c += cnt[i];
                if (c == 0 || rightCnt[i + 1] == 0) continue;
                float cost = acc.area() * c + rightArea[i + 1] * rightCnt[i + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = i; }
            }
        }
        const float parentArea = box.area();
        const float leafCost = C_ISECT * (float) n;
        size_t mid;
        if (bestAxis < 0) {
            if (n <= 8) return makeLeaf(b, e);
            mid = b + n / 2;   /* all centroids coincide: split in the middle */
        } else {
            float splitCost = C_TRAV + C_ISECT * bestCost / (parentArea > 0 ? parentArea : 1.0f);
            if (n <= MAX_LEAF && splitCost >= leafCost) return makeLeaf(b, e);
            float lo = cb.mn[bestAxis], hi = cb.mx[bestAxis];
            const float scale = NBINS / (hi - lo);
            BuildTri *first = &T[b], *last = &T[e];
            BuildTri *m = std::partition(first, last, [&](const BuildTri &t) {
                int k = (int) ((t.c[bestAxis] - lo) * scale); k = std::min(std::max(k, 0), NBINS - 1);
                return k <= bestBin;
            });
            mid = b + (size_t) (m - first);
            if (mid == b || mid == e) mid = b + n / 2;
        }
        const uint32_t idx = out.nNodes2++;
        out.nodes2.resize((size_t) out.nNodes2 * 16);
        Box lb, rb;
        int32_t l = build(b, mid, lb, depth + 1);
        int32_t r = build(mid, e, rb, depth + 1);
        Box lp = lb, rp = rb; pad(lp); pad(rp);
        float *nd = &out.nodes2[(size_t) idx * 16];
        nd[0] = lp.mn[0]; nd[1] = lp.mn[1]; nd[2] = lp.mn[2]; nd[3] = lp.mx[0];
        nd[4] = lp.mx[1]; nd[5] = lp.mx[2]; nd[6] = rp.mn[0]; nd[7] = rp.mn[1];
        nd[8] = rp.mn[2]; nd[9] = rp.mx[0]; nd[10] = rp.mx[1]; nd[11] = rp.mx[2];
        nd[12] = bits2f((uint32_t) l); nd[13] = bits2f((uint32_t) r); nd[14] = 0; nd[15] = 0;
        return (int32_t) idx;
    }

    /* the same recursion over a reference LIST, with spatial splits: at every node the best object split (binned over the
       centroids, as above) competes with the best spatial split (SBINS slabs of the node box per axis, triangles clipped to the
       slabs they cross); a reference that straddles the chosen plane is split into two references with clipped boxes unless
       keeping it whole on one side is cheaper ("reference unsplitting", section 4.4 of the paper). */
    int32_t buildSpatial(std::vector<BuildTri> &refs, Box &box, uint32_t depth) {
        out.maxDepth = std::max(out.maxDepth, depth);
        box.reset();
        Box cb; cb.reset();
        for (const BuildTri &t : refs) { box.grow(t.bmin, t.bmax); cb.growPt(t.c); }
        const size_t n = refs.size();
        if (n == 1) return makeLeaf(refs.data(), refs.data() + n);

        /* object split */
        float bestCost = INFINITY; int bestAxis = -1, bestBin = -1; Box bestL, bestR; bestL.reset(); bestR.reset();
        for (int axis = 0; axis < 3; ++axis) {
            float lo = cb.mn[axis], hi = cb.mx[axis];
            if (!(hi > lo)) continue;
            Box bins[NBINS]; uint32_t cnt[NBINS];
            for (int i = 0; i < NBINS; ++i) { bins[i].reset(); cnt[i] = 0; }
            const float scale = NBINS / (hi - lo);
            for (const BuildTri &t : refs) {
                int k = (int) ((t.c[axis] - lo) * scale); k = std::min(std::max(k, 0), NBINS - 1);
                bins[k].grow(t.bmin, t.bmax); cnt[k]++;
            }
            Box rightBox[NBINS]; uint32_t rightCnt[NBINS];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int i = NBINS - 1; i > 0; --i) { acc.grow(bins[i].mn, bins[i].mx); c += cnt[i]; rightBox[i] = acc; rightCnt[i] = c; }
            acc.reset(); c = 0;
            for (int i = 0; i < NBINS - 1; ++i) {
                acc.grow(bins[i].mn, bins[i].mx); c += cnt[i];
                if (c == 0 || rightCnt[i + 1] == 0) continue;
                float cost = acc.area() * c + rightBox[i + 1].area() * rightCnt[i + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = i; bestL = acc; bestR = rightBox[i + 1]; }
            }
        }
        /* spatial split, when the object split's children overlap enough */
        float sCost = INFINITY; int sAxis = -1; float sPos = 0;
        bool trySpatial = depth < 48 && slack > 0;               /* (the budget itself is enforced when a split is accepted) */
        if (trySpatial && bestAxis >= 0) {
            Box ov; for (int a = 0; a < 3; ++a) { ov.mn[a] = std::max(bestL.mn[a], bestR.mn[a]); ov.mx[a] = std::min(bestL.mx[a], bestR.mx[a]); }
            trySpatial = ov.area() / rootArea > alpha;
        }
        if (trySpatial) {
            for (int axis = 0; axis < 3; ++axis) {
                const float lo = box.mn[axis], hi = box.mx[axis];
                if (!(hi > lo)) continue;
                Box bins[SBINS]; uint32_t enter[SBINS], leave[SBINS];
                for (int i = 0; i < SBINS; ++i) { bins[i].reset(); enter[i] = leave[i] = 0; }
                const double w = ((double) hi - lo) / SBINS;
                for (const BuildTri &t : refs) {
                    int b0 = (int) (((double) t.bmin[axis] - lo) / w), b1 = (int) (((double) t.bmax[axis] - lo) / w);
                    b0 = std::min(std::max(b0, 0), SBINS - 1); b1 = std::min(std::max(b1, b0), SBINS - 1);
                    Box rb; for (int a = 0; a < 3; ++a) { rb.mn[a] = t.bmin[a]; rb.mx[a] = t.bmax[a]; }
                    if (b0 == b1) bins[b0].grow(rb.mn, rb.mx);
                    else for (int b = b0; b <= b1; ++b) {
                        Box cbx;
                        if (clippedBounds(t.prim, axis, lo + b * w, lo + (b + 1) * w, rb, cbx)) bins[b].grow(cbx.mn, cbx.mx);
                    }
                    enter[b0]++; leave[b1]++;
                }
                float rightArea[SBINS]; uint32_t rightCnt[SBINS];
                Box acc; acc.reset(); uint32_t c = 0;
                for (int i = SBINS - 1; i > 0; --i) { acc.grow(bins[i].mn, bins[i].mx); c += leave[i]; rightArea[i] = acc.area(); rightCnt[i] = c; }
                acc.reset(); c = 0;
                for (int i = 0; i < SBINS - 1; ++i) {
                    acc.grow(bins[i].mn, bins[i].mx); c += enter[i];
                    if (c == 0 || rightCnt[i + 1] == 0) continue;
                    const float cost = acc.area() * c + rightArea[i + 1] * rightCnt[i + 1];
                    if (cost < sCost) { sCost = cost; sAxis = axis; sPos = (float) (lo + (i + 1) * w); }
                }
            }
        }
        const float parentArea = box.area();
        const float leafCost = C_ISECT * (float) n;
        std::vector<BuildTri> L, R;
        bool done = false;
        if (sAxis >= 0 && sCost < bestCost) {
            /* partition by the plane; straddlers are split or kept whole on one side */
            Box lb, rb; lb.reset(); rb.reset();
            std::vector<const BuildTri *> straddle;
            for (const BuildTri &t : refs) {
                if (t.bmax[sAxis] <= sPos) { L.push_back(t); lb.grow(t.bmin, t.bmax); }
                else if (t.bmin[sAxis] >= sPos) { R.push_back(t); rb.grow(t.bmin, t.bmax); }
                else straddle.push_back(&t);
            }
            size_t nl = L.size() + straddle.size(), nr = R.size() + straddle.size();
            struct Piece { Box l, r; bool hasL, hasR; };
            std::vector<Piece> pieces(straddle.size());
            for (size_t i = 0; i < straddle.size(); ++i) {
                const BuildTri &t = *straddle[i];
                Box tb; for (int a = 0; a < 3; ++a) { tb.mn[a] = t.bmin[a]; tb.mx[a] = t.bmax[a]; }
                pieces[i].hasL = clippedBounds(t.prim, sAxis, -INFINITY, sPos, tb, pieces[i].l);
                pieces[i].hasR = clippedBounds(t.prim, sAxis, sPos, INFINITY, tb, pieces[i].r);
                if (pieces[i].hasL) lb.grow(pieces[i].l.mn, pieces[i].l.mx);
                if (pieces[i].hasR) rb.grow(pieces[i].r.mn, pieces[i].r.mx);
            }
            for (size_t i = 0; i < straddle.size(); ++i) {
                const BuildTri &t = *straddle[i];
                const Piece &pc = pieces[i];
                if (!pc.hasL && !pc.hasR) { L.push_back(t); continue; }                 /* (numerically empty: keep it somewhere) */
                if (!pc.hasR) { L.push_back(makeRef(pc.l, t.prim)); --nr; continue; }
                if (!pc.hasL) { R.push_back(makeRef(pc.r, t.prim)); --nl; continue; }
                Box lw = lb, rw = rb; lw.grow(t.bmin, t.bmax); rw.grow(t.bmin, t.bmax);
                const float cSplit = lb.area() * nl + rb.area() * nr;
                const float cLeft = lw.area() * nl + rb.area() * (nr - 1), cRight = lb.area() * (nl - 1) + rw.area() * nr;
                if (cLeft < cSplit && cLeft <= cRight) { L.push_back(t); lb = lw; --nr; }
                else if (cRight < cSplit) { R.push_back(t); rb = rw; --nl; }
                else { L.push_back(makeRef(pc.l, t.prim)); R.push_back(makeRef(pc.r, t.prim)); }
            }
            if (!L.empty() && !R.empty() && L.size() < n && R.size() < n && L.size() + R.size() - n <= slack) { done = true; slack -= L.size() + R.size() - n; }
            else { L.clear(); R.clear(); }
        }
        if (!done) {
            if (bestAxis < 0) {
                if (n <= 8) return makeLeaf(refs.data(), refs.data() + n);
                L.assign(refs.begin(), refs.begin() + n / 2); R.assign(refs.begin() + n / 2, refs.end());
            } else {
                const float splitCost = C_TRAV + C_ISECT * bestCost / (parentArea > 0 ? parentArea : 1.0f);
                if (n <= (size_t) MAX_LEAF && splitCost >= leafCost) return makeLeaf(refs.data(), refs.data() + n);
                const float lo = cb.mn[bestAxis], hi = cb.mx[bestAxis];
                const float scale = NBINS / (hi - lo);
                for (const BuildTri &t : refs) {
                    int k = (int) ((t.c[bestAxis] - lo) * scale); k = std::min(std::max(k, 0), NBINS - 1);
                    (k <= bestBin ? L : R).push_back(t);
                }
                if (L.empty() || R.empty()) { L.assign(refs.begin(), refs.begin() + n / 2); R.assign(refs.begin() + n / 2, refs.end()); }
            }
        } else if (n <= (size_t) MAX_LEAF) {
            const float splitCost = C_TRAV + C_ISECT * sCost / (parentArea > 0 ? parentArea : 1.0f);
            if (splitCost >= leafCost) { slack += L.size() + R.size() - n; return makeLeaf(refs.data(), refs.data() + n); }
        }
        std::vector<BuildTri>().swap(refs);
        const uint32_t idx = out.nNodes2++;
        out.nodes2.resize((size_t) out.nNodes2 * 16);
        Box lbx, rbx;
        int32_t l, r;
        const bool deal = dealLevels > 0 && L.size() >= PAR_MIN_REFS && R.size() >= PAR_MIN_REFS;
        if (deal && parallelLevels > 0) {
            /* the right subtree on another thread, into arrays of its own; spliced in behind the left subtree afterwards (node and
               record indices shifted), so the layout is the depth-first one of the serial build.  The reference budget is dealt in
               proportion to the two sides: deterministic, whatever the threads' timing. */
            HostBVH sub; std::vector<BuildTri> none;
            Builder SB(none, sub, positions, indices);
            SB.MAX_LEAF = MAX_LEAF; SB.C_TRAV = C_TRAV; SB.C_ISECT = C_ISECT; SB.spatial = spatial; SB.alpha = alpha; SB.rootArea = rootArea;
            for (int a = 0; a < 3; ++a) SB.extent[a] = extent[a];
            SB.parallelLevels = parallelLevels - 1; SB.dealLevels = dealLevels - 1;
            const size_t total = slack, slackR = (size_t) ((double) total * (double) R.size() / (double) (L.size() + R.size()));
            SB.slack = slackR; slack = total - slackR;
            int32_t rr = 0;
            std::exception_ptr failed;                            /* (an exception that leaves a std::thread is std::terminate) */
            std::thread worker([&]() { try { rr = SB.buildSpatial(R, rbx, depth + 1); } catch (...) { failed = std::current_exception(); } });
            const int saved = parallelLevels, savedDeal = dealLevels; parallelLevels = saved - 1; dealLevels = savedDeal - 1;
            try { l = buildSpatial(L, lbx, depth + 1); } catch (...) { worker.join(); throw; }
            parallelLevels = saved; dealLevels = savedDeal;
            worker.join();
            if (failed) std::rethrow_exception(failed);
            slack += SB.slack;
            const uint32_t nodeOff = out.nNodes2, recOff = (uint32_t) (out.tris.size() / 12);
            auto shift = [&](int32_t ref) -> int32_t {
                if (ref >= 0) return ref + (int32_t) nodeOff;
                const uint32_t q = ~(uint32_t) ref;
                return ~(int32_t) ((((q >> 3) + recOff) << 3) | (q & 7u));
            };
            for (uint32_t i = 0; i < sub.nNodes2; ++i) {
                float *nd2 = &sub.nodes2[(size_t) i * 16];
                uint32_t a, b; memcpy(&a, &nd2[12], 4); memcpy(&b, &nd2[13], 4);
                a = (uint32_t) shift((int32_t) a); b = (uint32_t) shift((int32_t) b);
                memcpy(&nd2[12], &a, 4); memcpy(&nd2[13], &b, 4);
            }
            out.nodes2.insert(out.nodes2.end(), sub.nodes2.begin(), sub.nodes2.begin() + (size_t) sub.nNodes2 * 16);
            out.nNodes2 += sub.nNodes2;
            out.tris.insert(out.tris.end(), sub.tris.begin(), sub.tris.end());
            out.nLeaves += sub.nLeaves; out.nTriRefs += sub.nTriRefs; out.maxDepth = std::max(out.maxDepth, sub.maxDepth);
            r = shift(rr);
        } else if (deal) {
            /* the same deal of the reference budget without a second thread: the tree does not depend on the host's core count */
            const size_t total = slack, slackR = (size_t) ((double) total * (double) R.size() / (double) (L.size() + R.size()));
            const int savedDeal = dealLevels; dealLevels = savedDeal - 1;
            slack = total - slackR;
            l = buildSpatial(L, lbx, depth + 1);
            const size_t leftL = slack;
            slack = slackR;
            r = buildSpatial(R, rbx, depth + 1);
            slack += leftL;
            dealLevels = savedDeal;
        } else {
            l = buildSpatial(L, lbx, depth + 1);
            r = buildSpatial(R, rbx, depth + 1);
        }
        Box lp = lbx, rp = rbx; pad(lp); pad(rp);
        float *nd = &out.nodes2[(size_t) idx * 16];
        nd[0] = lp.mn[0]; nd[1] = lp.mn[1]; nd[2] = lp.mn[2]; nd[3] = lp.mx[0];
        nd[4] = lp.mx[1]; nd[5] = lp.mx[2]; nd[6] = rp.mn[0]; nd[7] = rp.mn[1];
        nd[8] = rp.mn[2]; nd[9] = rp.mx[0]; nd[10] = rp.mx[1]; nd[11] = rp.mx[2];
        nd[12] = bits2f((uint32_t) l); nd[13] = bits2f((uint32_t) r); nd[14] = 0; nd[15] = 0;
        return (int32_t) idx;
    }
};

} // namespace detail

/*
 * Insertion-based optimisation of the binary tree (Bittner, Hapala, Havran: "Fast Insertion-Based Optimization of Bounding Volume
 * Hierarchies", CGF 2013).  One pass for scenes of SPATIAL_MIN_TRIS triangles or more (PHIP_BVH_OPT=<passes> overrides; round 2: measured on
 * the CPU twin of the traversal, tools/bvh_quality.py: node steps per ray -7 % / -13 % on the atrium, -5 % / -9 % on the glass room).  One step: take an inner node n out of the tree (its parent goes with it, the sibling moves
 * up), then put each of n's two subtrees back where it costs least -- branch-and-bound over the whole tree for the node Y that
 * minimises area(Y u X) + the growth of Y's ancestors -- under a new parent made from one of the two freed nodes.  Leaves (record
 * ranges) are never touched, so the records and the answers stay what they were; only the inner topology changes.
 */
namespace detail {
struct Reinserter {
    struct Inner { int32_t parent; int32_t child[2]; Box box[2]; };
    std::vector<Inner> n;                      /* inner nodes; child >= 0: inner, < 0: leaf reference */
    std::vector<int32_t> leafParent;           /* by first record of the leaf */
    int32_t root;
    static uint32_t leafKey(int32_t ref) { return (~(uint32_t) ref) >> 3; }
    Box boxOf(int32_t i) const { Box b = n[i].box[0]; b.grow(n[i].box[1].mn, n[i].box[1].mx); return b; }
    int32_t parentOf(int32_t ref) const { return ref >= 0 ? n[ref].parent : leafParent[leafKey(ref)]; }
    void setParent(int32_t ref, int32_t p) { if (ref >= 0) n[ref].parent = p; else leafParent[leafKey(ref)] = p; }
    int slotOf(int32_t p, int32_t ref) const { return n[p].child[0] == ref ? 0 : 1; }
    static float unionArea(const Box &a, const Box &b) { Box u = a; u.grow(b.mn, b.mx); return u.area(); }
    /* boxes of the ancestors of p's slots, from p up to the root */
    void refit(int32_t p) {
        while (p >= 0) {
            const int32_t g = n[p].parent;
            if (g < 0) break;
            n[g].box[slotOf(g, p)] = boxOf(p);
            p = g;
        }
    }
    /* best node to pair subtree X (box bx) with */
    struct Cand { float induced; int32_t ref; Box box; };
    mutable std::vector<Cand> heap;            /* (kept between searches: no allocation per insertion) */
    uint32_t maxPops = 512;                    /* search bound per insertion: with many identical boxes the branch-and-bound prunes nothing and one
                                                  search visits the whole tree -- O(n^2) per pass (round-2 advice: 1 M coincident triangles took 121 s);
                                                  the best candidate found within the bound is a valid place all the same */
    int32_t findBest(const Box &bx, Box &bestBox) const {
        auto cmp = [](const Cand &a, const Cand &b) { return a.induced > b.induced; };
        heap.clear();
        const float ax = bx.area();
        float best = INFINITY; int32_t bestRef = root; bestBox = boxOf(root);
        heap.push_back({ 0.0f, root, boxOf(root) });
        uint32_t pops = 0;
        while (!heap.empty() && pops++ < maxPops) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            const Cand c = heap.back(); heap.pop_back();
            if (c.induced + ax >= best) break;                           /* every remaining candidate is at least this bad */
            const float direct = unionArea(c.box, bx);
            const float total = c.induced + direct;
            if (total < best) { best = total; bestRef = c.ref; bestBox = c.box; }
            if (c.ref >= 0) {
                const float ind = total - c.box.area();                  /* growth this node would suffer as an ancestor */
                if (ind + ax < best)
                    for (int k = 0; k < 2; ++k) { heap.push_back({ ind, n[c.ref].child[k], n[c.ref].box[k] }); std::push_heap(heap.begin(), heap.end(), cmp); }
            }
        }
        return bestRef;
    }
    /* pair subtree x with node y under the free inner node q */
    void insert(int32_t x, const Box &bx, int32_t y, const Box &by, int32_t q) {
        const int32_t p = parentOf(y);
        n[q].child[0] = y; n[q].box[0] = by; n[q].child[1] = x; n[q].box[1] = bx; n[q].parent = p;
        if (p >= 0) { const int sl = slotOf(p, y); n[p].child[sl] = q; }
        else root = q;
        setParent(y, q); setParent(x, q);
        if (p >= 0) { n[p].box[slotOf(p, q)] = boxOf(q); refit(p); }
    }
    double sah() const {
        double c = 0;
        for (size_t i = 0; i < n.size(); ++i) c += n[i].box[0].area() + n[i].box[1].area();
        return c;
    }
    /* one pass over the inner nodes, largest boxes first */
    void pass() {
        std::vector<std::pair<float, int32_t>> order;
        for (int32_t i = 0; i < (int32_t) n.size(); ++i) order.push_back({ -boxOf(i).area(), i });
        std::sort(order.begin(), order.end());
        for (const auto &o : order) {
            const int32_t v = o.second;
            const int32_t p = n[v].parent;
            if (v == root || p < 0 || p == root) continue;               /* keep the top two levels in place */
            const int32_t g = n[p].parent;
            const int sv = slotOf(p, v);
            const int32_t sib = n[p].child[1 - sv]; const Box sibBox = n[p].box[1 - sv];
            /* take v and p out: the sibling moves up into p's slot */
            const int sp = slotOf(g, p);
            n[g].child[sp] = sib; n[g].box[sp] = sibBox; setParent(sib, g);
            refit(g);
            const int32_t kids[2] = { n[v].child[0], n[v].child[1] }; const Box kb[2] = { n[v].box[0], n[v].box[1] };
            const int first = kb[0].area() >= kb[1].area() ? 0 : 1;     /* the larger subtree first */
            const int32_t freeIds[2] = { v, p };
            for (int t = 0; t < 2; ++t) {
                const int k = t == 0 ? first : 1 - first;
                Box by; const int32_t y = findBest(kb[k], by);
                insert(kids[k], kb[k], y, by, freeIds[t]);
            }
        }
    }
};
} // namespace detail

/*
 * Compressed wide BVH (CWBVH, Ylitie et al. 2017), built by collapsing the binary SAH tree.  The ray kernels of the big scenes
 * are bound by the CU's vector-memory path (every lane of a wave fetches its own node: 64 cache lines per load instruction),
 * so what counts is bytes and dependent round trips per ray: one 80-byte node replaces ~2.3 of the 128-byte BVH4 nodes.
 *
 * Node = 5 x 16 bytes:
 *   [0] p.x p.y p.z (float: the node box's lower corner)   e.x | e.y << 8 | e.z << 16 | imask << 24
 *       (e = biased exponent byte of the per-axis power-of-two grid step; imask bit s: slot s holds an inner node)
 *   [1] childBase  triBase  meta[0..3]  meta[4..7]
 *       inner children are nodes childBase + rank (rank = number of inner slots below s); the node's leaf triangles are
 *       records triBase + offset.  meta byte of slot s: 0 = empty; inner: 0x20 | (24 + s); leaf: unary(count) << 5 | offset
 *       with count 1..3 -> 0b001, 0b011, 0b111 and offset < 24
 *   [2] qlo.x[0..3] qlo.x[4..7] qlo.y[0..3] qlo.y[4..7]    [3] qlo.z[0..3] qlo.z[4..7] qhi.x[0..3] qhi.x[4..7]
 *   [4] qhi.y[0..3] qhi.y[4..7] qhi.z[0..3] qhi.z[4..7]    child box = p + q * 2^(e - 127), rounded outwards (conservative)
 * Children sit in the slot whose octant direction ((s & 1 ? + : -), (s & 2 ? + : -), (s & 4 ? + : -)) best matches their
 * offset from the node centre (greedy assignment), so that slot ^ rayOctant orders them approximately front to back
 * without a sort.  Empty slots have qlo = 255 > qhi = 0.
 */
template <typename Children2>
inline void buildWide(HostBVH &out, int32_t root2, const detail::Box &rootBox, Children2 children2) {
    out.wnodes.clear(); out.wtris.clear(); out.nWNodes = 0; out.wMaxDepth = 0; out.nWTris = 0; out.wSahCost = 0;
    /* (root2 < 0: the whole scene is ONE leaf -- round 6: it still gets a wide root whose only children are the pieces of that leaf, because every scene's ray kernels walk
       the wide tree; the collapse below has nothing to decide then) */
    typedef detail::ChildRef Child;
    struct WChild { Child c; uint32_t firstTri, nTris; };    /* leaf pieces carry their record range (in out.tris) */
    struct Item { int32_t ref2; uint32_t index, depth; detail::Box box; };
    /* ---- optimal collapse (Ylitie et al. 2017, section 3.1): C[n][i] = the least SAH cost of representing BVH2 subtree n with
       at most i slots of its wide parent: either n becomes ONE wide node (cost area * cNode + the best distribution of its own
       8 slots over its two children) or its slots are split between its children.  A leaf of P records needs ceil(P / 3)
       slots.  Filled bottom-up; the tree is then emitted top-down along the recorded decisions.  The greedy "open the
       largest child" collapse it replaces left the bottom of the tree underfilled (4 children per node on average). ---- */
    const float cNode = 2.5f, cTri = 1.0f;                   /* a node step costs about 2.5 Wald tests in k_rays_w */
    const uint32_t n2 = out.nNodes2;
    std::vector<float> C((size_t) n2 * 8, INFINITY);          /* C[n * 8 + i], i = 1..7; [n * 8 + 0] = cost as a wide node's root */
    std::vector<uint8_t> split((size_t) n2 * 9, 0);           /* split[n * 9 + j]: slots given to the LEFT child when n distributes j slots (j = 2..8) */
    auto leafCost = [&](const Child &c, int i) -> float {     /* c.ref < 0 */
        const uint32_t cnt = ((~(uint32_t) c.ref) & 7u) + 1u;
        return i >= (int) ((cnt + 2) / 3) ? c.box.area() * cTri * (float) cnt : INFINITY;
    };
    auto costOf = [&](const Child &c, int i) -> float { return c.ref < 0 ? leafCost(c, i) : C[(size_t) c.ref * 8 + std::min(i, 7)]; };
    {
        /* post-order over the BVH2 (children have larger indices than their parent in build order: iterate backwards) */
        for (int32_t n = (int32_t) n2 - 1; n >= 0; --n) {
            Child two[2]; children2(n, two);
            float dist[9];
            for (int j = 2; j <= 8; ++j) {
                float best = INFINITY; int bk = 1;
                for (int k = 1; k < j; ++k) {
                    const float v = costOf(two[0], k) + costOf(two[1], j - k);
                    if (v < best) { best = v; bk = k; }
                }
                dist[j] = best; split[(size_t) n * 9 + j] = (uint8_t) bk;
            }
            detail::Box nb; nb.reset(); nb.grow(two[0].box.mn, two[0].box.mx); nb.grow(two[1].box.mn, two[1].box.mx);
            float *c = &C[(size_t) n * 8];
            c[1] = nb.area() * cNode + dist[8];
            for (int i = 2; i <= 7; ++i) c[i] = std::min(dist[i], c[i - 1]);
        }
    }
    /* children of BVH2 node n when it distributes j slots */
    std::function<void(int32_t, int, std::vector<Child> &)> expand = [&](int32_t n, int j, std::vector<Child> &outc) {
        Child two[2]; children2(n, two);
        const int k = split[(size_t) n * 9 + j];
        const int budget[2] = { k, j - k };
        for (int s2 = 0; s2 < 2; ++s2) {
            const Child &c = two[s2];
            if (c.ref < 0) { outc.push_back(c); continue; }
            int i = std::min(budget[s2], 7);
            const float *cc = &C[(size_t) c.ref * 8];
            while (i > 1 && cc[i] == cc[i - 1]) --i;
            if (i == 1) outc.push_back(c);                    /* c becomes a wide node of its own */
            else expand(c.ref, i, outc);
        }
    };
    std::vector<Item> queue;
    queue.push_back({ root2, 0u, 1u, rootBox });
    out.nWNodes = 1;
    double cost = 0; const double rootA = rootBox.area() > 0 ? rootBox.area() : 1.0;
    for (size_t head = 0; head < queue.size(); ++head) {
        const Item it = queue[head];
        out.wMaxDepth = std::max(out.wMaxDepth, it.depth);
        /* the children of this wide node, as the SAH-optimal collapse (cost table below) distributes its 8 slots */
        std::vector<Child> ch;
        if (it.ref2 < 0) ch.push_back(Child{ it.ref2, it.box }); else expand(it.ref2, 8, ch);
        /* leaf children of more than 3 records are split into pieces of <= 3 (same box) */
        std::vector<WChild> wc;
        for (const Child &c : ch) {
            if (c.ref >= 0) { wc.push_back({ c, 0u, 0u }); continue; }
            const uint32_t r = ~(uint32_t) c.ref, first = r >> 3, cnt = (r & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; k += 3) wc.push_back({ c, first + k, std::min(3u, cnt - k) });
        }
        /* node box = union of the (padded) child boxes */
        detail::Box nb; nb.reset();
        for (const WChild &w : wc) nb.grow(w.c.box.mn, w.c.box.mx);
        float p[3]; uint32_t e8[3]; double step[3];
        for (int a = 0; a < 3; ++a) {
            p[a] = nb.mn[a];
            const double ext = (double) nb.mx[a] - (double) p[a];
            int e = -120;
            if (ext > 0) { int ex; std::frexp(ext / 255.0, &ex); e = ex; }            /* 2^e >= ext / 255 */
            while (std::ldexp(1.0, e) * 255.0 < ext) ++e;
            e = std::min(127, std::max(-120, e));
            e8[a] = (uint32_t) (e + 127); step[a] = std::ldexp(1.0, e);
        }
        /* slot assignment by octant direction (greedy on the dot product) */
        const float cen[3] = { 0.5f * (nb.mn[0] + nb.mx[0]), 0.5f * (nb.mn[1] + nb.mx[1]), 0.5f * (nb.mn[2] + nb.mx[2]) };
        int slotOf[8]; bool slotUsed[8] = { false }, childDone[8] = { false };
        for (size_t k = 0; k < wc.size(); ++k) {
            float bestV = -INFINITY; int bc = -1, bs = -1;
            for (size_t c = 0; c < wc.size(); ++c) {
                if (childDone[c]) continue;
                const float d[3] = { 0.5f * (wc[c].c.box.mn[0] + wc[c].c.box.mx[0]) - cen[0], 0.5f * (wc[c].c.box.mn[1] + wc[c].c.box.mx[1]) - cen[1],
                                     0.5f * (wc[c].c.box.mn[2] + wc[c].c.box.mx[2]) - cen[2] };
                for (int sl = 0; sl < 8; ++sl) {
                    if (slotUsed[sl]) continue;
                    const float v = ((sl & 1) ? d[0] : -d[0]) + ((sl & 2) ? d[1] : -d[1]) + ((sl & 4) ? d[2] : -d[2]);
                    if (v > bestV) { bestV = v; bc = (int) c; bs = sl; }
                }
            }
            childDone[bc] = true; slotUsed[bs] = true; slotOf[bc] = bs;
        }
        int childAt[8]; for (int sl = 0; sl < 8; ++sl) childAt[sl] = -1;
        for (size_t c = 0; c < wc.size(); ++c) childAt[slotOf[c]] = (int) c;
        /* emit */
        uint32_t nInner = 0; for (const WChild &w : wc) if (w.c.ref >= 0) ++nInner;
        const uint32_t childBase = out.nWNodes; out.nWNodes += nInner;
        const uint32_t triBase = out.nWTris;
        if (out.wnodes.size() < (size_t) out.nWNodes * 20) out.wnodes.resize((size_t) out.nWNodes * 20, 0u);
        uint32_t *nd = &out.wnodes[(size_t) it.index * 20];
        uint8_t meta[8], q[6][8]; uint32_t imask = 0, rank = 0, triOff = 0;
        for (int sl = 0; sl < 8; ++sl) {
            meta[sl] = 0;
            for (int a = 0; a < 3; ++a) { q[a][sl] = 255; q[3 + a][sl] = 0; }
            const int c = childAt[sl];
            if (c < 0) continue;
            const WChild &w = wc[c];
            for (int a = 0; a < 3; ++a) {
                const double lo = std::floor(((double) w.c.box.mn[a] - (double) p[a]) / step[a]);
                const double hi = std::ceil(((double) w.c.box.mx[a] - (double) p[a]) / step[a]);
                q[a][sl] = (uint8_t) std::min(255.0, std::max(0.0, lo));
                q[3 + a][sl] = (uint8_t) std::min(255.0, std::max(0.0, hi));
            }
            cost += w.c.box.area() / rootA * (w.c.ref < 0 ? (double) w.nTris : 1.0);
            if (w.c.ref >= 0) {
                imask |= 1u << sl;
                meta[sl] = (uint8_t) (0x20u | (24u + (uint32_t) sl));
                queue.push_back({ w.c.ref, childBase + rank, it.depth + 1, w.c.box });
                ++rank;
            } else {
                meta[sl] = (uint8_t) ((((1u << w.nTris) - 1u) << 5) | triOff);
                out.wtris.insert(out.wtris.end(), out.tris.begin() + (size_t) w.firstTri * 12, out.tris.begin() + (size_t) (w.firstTri + w.nTris) * 12);
                triOff += w.nTris; out.nWTris += w.nTris;
            }
        }
        auto pack4 = [](const uint8_t *b) { return (uint32_t) b[0] | ((uint32_t) b[1] << 8) | ((uint32_t) b[2] << 16) | ((uint32_t) b[3] << 24); };
        memcpy(&nd[0], &p[0], 4); memcpy(&nd[1], &p[1], 4); memcpy(&nd[2], &p[2], 4);
        nd[3] = e8[0] | (e8[1] << 8) | (e8[2] << 16) | (imask << 24);
        nd[4] = childBase; nd[5] = triBase; nd[6] = pack4(meta); nd[7] = pack4(meta + 4);
        nd[8] = pack4(q[0]); nd[9] = pack4(q[0] + 4); nd[10] = pack4(q[1]); nd[11] = pack4(q[1] + 4);
        nd[12] = pack4(q[2]); nd[13] = pack4(q[2] + 4); nd[14] = pack4(q[3]); nd[15] = pack4(q[3] + 4);
        nd[16] = pack4(q[4]); nd[17] = pack4(q[4] + 4); nd[18] = pack4(q[5]); nd[19] = pack4(q[5] + 4);
    }
    out.wSahCost = (float) (cost + 1.0);
}

inline void buildBVH(const float *positions, const uint32_t *indices, uint32_t nTris, HostBVH &out, const float *cameraPosition = nullptr) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<BuildTri> T; T.reserve(nTris);
    detail::Box tight; tight.reset();
    for (uint32_t i = 0; i < nTris; ++i) {
        BuildTri bt; bt.prim = i;
        for (int a = 0; a < 3; ++a) { bt.bmin[a] = INFINITY; bt.bmax[a] = -INFINITY; }
        for (int v = 0; v < 3; ++v) {
            const float *p = positions + 3 * (size_t) indices[3 * (size_t) i + v];
            for (int a = 0; a < 3; ++a) { bt.bmin[a] = std::min(bt.bmin[a], p[a]); bt.bmax[a] = std::max(bt.bmax[a], p[a]); }
        }
        for (int a = 0; a < 3; ++a) bt.c[a] = 0.5f * (bt.bmin[a] + bt.bmax[a]);
        tight.grow(bt.bmin, bt.bmax);
        T.push_back(bt);
    }
    out = HostBVH();
    out.nodes2.reserve((size_t) nTris * 16);
    out.tris.reserve((size_t) nTris * 12);
    for (int a = 0; a < 3; ++a) { out.tightMin[a] = tight.mn[a]; out.tightMax[a] = tight.mx[a]; }
    /* scene box: tight box enlarged exactly like the reference's kd-tree root (gkdtree.h:1213-1220):
       min -= (max-min)*eps + eps; max += (max-min_new)*eps + eps */
    const float eps = 1e-3f;
    for (int a = 0; a < 3; ++a) {
        float mn = tight.mn[a], mx = tight.mx[a];
        mn -= (mx - mn) * eps + eps;
        mx += (mx - mn) * eps + eps;
        out.sceneMin[a] = mn; out.sceneMax[a] = mx;
    }
    if (nTris == 0) {
        /* empty scene: a single never-hit leaf */
        float rec[12] = { 0 }; rec[0] = detail::bits2f(3u); rec[10] = detail::bits2f(0xFFFFFFFFu);
        out.tris.assign(rec, rec + 12);
        out.rootRef = ~(int32_t) 0;
        for (int a = 0; a < 3; ++a) { out.sceneMin[a] = out.tightMin[a] = 0; out.sceneMax[a] = out.tightMax[a] = 0; }
        return;
    }
    detail::Builder B(T, out, positions, indices);
    for (int a = 0; a < 3; ++a) B.extent[a] = tight.mx[a] - tight.mn[a];
    if (cameraPosition) {
        /* The camera's rays start outside the scene: with the origin 3000 scene extents away a distance is quantised to an eighth of a scene unit, and the Wald test -- the
           arbiter: the oracle's sweep over all triangles -- then accepts rays that pass a triangle's silhouette by up to ~4 ulp(|o|) ON ANY AXIS (the plane equation's numerator
           is rounded at the magnitude of the origin's LARGEST component and divided by the dominant component of the normal; round 6 found one such sample in 32768: the ray
           0.03 units beside the tall block of the Cornell box).  So every axis is padded by the largest component, not by its own (8 ulp). */
        const float m = std::max(std::fabs(cameraPosition[0]), std::max(std::fabs(cameraPosition[1]), std::fabs(cameraPosition[2])));
        for (int a = 0; a < 3; ++a) B.camPad[a] = 4.8e-7f * m;
    }
    detail::Box rootBox;
    /* spatial splits for the scenes that use the wide tree (small scenes are laid out for LDS record by record) */
    const char *sp = getenv("PHIP_BVH_SPATIAL");
    B.spatial = sp ? atoi(sp) != 0 : nTris >= SPATIAL_MIN_TRIS;
    int32_t root2;
    if (B.spatial) {
        B.rootArea = tight.area() > 0 ? tight.area() : 1.0;
        B.slack = (size_t) nTris / 2 + 64;                                   /* at most 1.5 references per triangle */
        {   /* the top levels' subtrees are built on 2^levels threads (PHIP_BVH_THREADS=1: serial) */
            unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1;
            if (const char *e = getenv("PHIP_BVH_THREADS")) hw = (unsigned) std::max(1, atoi(e));
            B.parallelLevels = 0; while ((2u << B.parallelLevels) <= hw && B.parallelLevels < 5) ++B.parallelLevels;
        }
        out.nodes2.reserve((size_t) nTris * 24); out.tris.reserve((size_t) nTris * 18);
        root2 = B.buildSpatial(T, rootBox, 1);
    } else
        root2 = B.build(0, T.size(), rootBox, 1);
    {
        /* one pass for the scenes that use the wide tree (the gate of the spatial splits); PHIP_BVH_OPT=<passes> overrides, 0 disables */
        const char *e = getenv("PHIP_BVH_OPT");
        const int passes = e ? atoi(e) : (nTris >= SPATIAL_MIN_TRIS ? 1 : 0);
        if (passes > 0 && root2 >= 0 && out.nNodes2 > 8) {
            const std::vector<float> nodes2Before = out.nodes2; const int32_t root2Before = root2;
            detail::Reinserter R;
            R.n.resize(out.nNodes2); R.leafParent.assign(out.tris.size() / 12 + 1, -1); R.root = root2;
            for (uint32_t i = 0; i < out.nNodes2; ++i) {
                const float *nd = &out.nodes2[(size_t) i * 16];
                detail::Box a, b;
                a.mn[0] = nd[0]; a.mn[1] = nd[1]; a.mn[2] = nd[2]; a.mx[0] = nd[3]; a.mx[1] = nd[4]; a.mx[2] = nd[5];
                b.mn[0] = nd[6]; b.mn[1] = nd[7]; b.mn[2] = nd[8]; b.mx[0] = nd[9]; b.mx[1] = nd[10]; b.mx[2] = nd[11];
                uint32_t l, r; memcpy(&l, &nd[12], 4); memcpy(&r, &nd[13], 4);
                R.n[i].box[0] = a; R.n[i].box[1] = b; R.n[i].child[0] = (int32_t) l; R.n[i].child[1] = (int32_t) r; R.n[i].parent = -1;
            }
            for (uint32_t i = 0; i < out.nNodes2; ++i)
                for (int k = 0; k < 2; ++k) R.setParent(R.n[i].child[k], (int32_t) i);
            R.n[root2].parent = -1;
            if (const char *pe = getenv("PHIP_BVH_OPT_POPS")) R.maxPops = (uint32_t) std::max(1, atoi(pe));
            const double before = R.sah();
            for (int it = 0; it < passes; ++it) R.pass();
            if (getenv("PHIP_DEBUG_TIMING")) fprintf(stderr, "phip: BVH reinsertion, %d passes: sum of child areas %.6g -> %.6g\n", passes, before, R.sah());
            /* write back, inner nodes renumbered in depth-first pre-order (a parent before its children: buildWide's post-order loop) */
            std::vector<float> nn((size_t) out.nNodes2 * 16);
            std::vector<int32_t> newId(out.nNodes2, -1), stack;
            uint32_t next = 0;
            stack.push_back(R.root);
            std::vector<int32_t> orderIds;
            while (!stack.empty()) {
                const int32_t v = stack.back(); stack.pop_back();
                newId[v] = (int32_t) next++; orderIds.push_back(v);
                for (int k = 1; k >= 0; --k) if (R.n[v].child[k] >= 0) stack.push_back(R.n[v].child[k]);
            }
            for (int32_t v : orderIds) {
                float *nd = &nn[(size_t) newId[v] * 16];
                const detail::Box &a = R.n[v].box[0], &b = R.n[v].box[1];
                nd[0] = a.mn[0]; nd[1] = a.mn[1]; nd[2] = a.mn[2]; nd[3] = a.mx[0]; nd[4] = a.mx[1]; nd[5] = a.mx[2];
                nd[6] = b.mn[0]; nd[7] = b.mn[1]; nd[8] = b.mn[2]; nd[9] = b.mx[0]; nd[10] = b.mx[1]; nd[11] = b.mx[2];
                const uint32_t l = (uint32_t) (R.n[v].child[0] >= 0 ? newId[R.n[v].child[0]] : R.n[v].child[0]);
                const uint32_t r = (uint32_t) (R.n[v].child[1] >= 0 ? newId[R.n[v].child[1]] : R.n[v].child[1]);
                memcpy(&nd[12], &l, 4); memcpy(&nd[13], &r, 4); nd[14] = 0; nd[15] = 0;
            }
            out.nodes2.swap(nn);
            root2 = 0;
            /* guard: reinsertion may chain nodes of identical boxes; a tree that got much deeper is not worth having (stack depth) */
            std::function<uint32_t(int32_t)> depthOf = [&](int32_t v) -> uint32_t {
                uint32_t l, r; memcpy(&l, &out.nodes2[(size_t) v * 16 + 12], 4); memcpy(&r, &out.nodes2[(size_t) v * 16 + 13], 4);
                return 1u + std::max((int32_t) l >= 0 ? depthOf((int32_t) l) : 1u, (int32_t) r >= 0 ? depthOf((int32_t) r) : 1u);
            };
            const uint32_t depthAfter = depthOf(root2);
            if (depthAfter > std::max<uint32_t>(out.maxDepth + 12u, 40u)) { out.nodes2 = nodes2Before; root2 = root2Before; }
        }
    }

    /* ---- collapse to BVH4 ---- */
    typedef detail::ChildRef Child;
    auto children2 = [&](int32_t ref, Child out2[2]) {
        const float *nd = &out.nodes2[(size_t) ref * 16];
        out2[0].box.mn[0] = nd[0]; out2[0].box.mn[1] = nd[1]; out2[0].box.mn[2] = nd[2]; out2[0].box.mx[0] = nd[3]; out2[0].box.mx[1] = nd[4]; out2[0].box.mx[2] = nd[5];
        out2[1].box.mn[0] = nd[6]; out2[1].box.mn[1] = nd[7]; out2[1].box.mn[2] = nd[8]; out2[1].box.mx[0] = nd[9]; out2[1].box.mx[1] = nd[10]; out2[1].box.mx[2] = nd[11];
        uint32_t l, r; memcpy(&l, &nd[12], 4); memcpy(&r, &nd[13], 4);
        out2[0].ref = (int32_t) l; out2[1].ref = (int32_t) r;
    };
    out.maxDepth = 0;
    double cost = 0; const double rootA = rootBox.area() > 0 ? rootBox.area() : 1.0;
    std::function<int32_t(int32_t, uint32_t)> collapse = [&](int32_t ref2, uint32_t depth) -> int32_t {
        out.maxDepth = std::max(out.maxDepth, depth);
        if (ref2 < 0) return ref2;                      /* leaf reference stays */
        Child ch[4]; int n = 2;
        children2(ref2, ch);
        while (n < 4) {                                 /* expand the inner child with the largest area */
            int best = -1; float bestA = -1;
            for (int i = 0; i < n; ++i) if (ch[i].ref >= 0 && ch[i].box.area() > bestA) { bestA = ch[i].box.area(); best = i; }
            if (best < 0) break;
            Child two[2]; children2(ch[best].ref, two);
            ch[best] = two[0]; ch[n++] = two[1];
        }
        const uint32_t idx = out.nNodes++;
        out.nodes.resize((size_t) out.nNodes * 32);
        int32_t refs[4];
        for (int i = 0; i < n; ++i) {
            refs[i] = collapse(ch[i].ref, depth + 1);
            cost += ch[i].box.area() / rootA * (ch[i].ref < 0 ? (double) (((~(uint32_t) ch[i].ref) & 7) + 1) : 1.0);
        }
        float *nd = &out.nodes[(size_t) idx * 32];
        for (int i = 0; i < 4; ++i) {
            const bool used = i < n;
            for (int a = 0; a < 3; ++a) {
                nd[4 * a + i] = used ? ch[i].box.mn[a] : INFINITY;
                nd[4 * (3 + a) + i] = used ? ch[i].box.mx[a] : -INFINITY;
            }
            nd[24 + i] = detail::bits2f(used ? (uint32_t) refs[i] : 0xffffffffu);   /* unused: never reached */
            nd[28 + i] = 0.0f;
        }
        return (int32_t) idx;
    };
    out.rootRef = collapse(root2, 1);
    out.sahCost = (float) (cost + 1.0);

    /* The traversal kernels stage nodes [0, K) in LDS: renumber so that the top of the tree comes first
       (breadth-first for the first TOP_BFS nodes, the rest keeps its depth-first order). */
    if (out.rootRef >= 0 && out.nNodes > 1) {
        const uint32_t TOP_BFS = 256;
        std::vector<uint32_t> order; order.reserve(out.nNodes);
        std::vector<uint8_t> taken(out.nNodes, 0);
        order.push_back((uint32_t) out.rootRef); taken[out.rootRef] = 1;
        for (size_t head = 0; head < order.size() && order.size() < TOP_BFS; ++head) {
            const float *nd = &out.nodes[(size_t) order[head] * 32];
            for (int i = 0; i < 4 && order.size() < TOP_BFS; ++i) {
                uint32_t r; memcpy(&r, &nd[24 + i], 4);
                if ((int32_t) r >= 0 && r < out.nNodes && !taken[r]) { taken[r] = 1; order.push_back(r); }
            }
        }
        for (uint32_t i = 0; i < out.nNodes; ++i) if (!taken[i]) order.push_back(i);
        std::vector<uint32_t> newIndex(out.nNodes);
        for (uint32_t i = 0; i < out.nNodes; ++i) newIndex[order[i]] = i;
        std::vector<float> nn(out.nodes.size());
        for (uint32_t i = 0; i < out.nNodes; ++i) {
            const float *src = &out.nodes[(size_t) order[i] * 32]; float *dst = &nn[(size_t) i * 32];
            memcpy(dst, src, 32 * sizeof(float));
            for (int c = 0; c < 4; ++c) {
                uint32_t r; memcpy(&r, &src[24 + c], 4);
                if ((int32_t) r >= 0 && r < out.nNodes && src[c] != INFINITY) { r = newIndex[r]; memcpy(&dst[24 + c], &r, 4); }
            }
        }
        out.nodes.swap(nn);
        out.rootRef = (int32_t) newIndex[out.rootRef];
    }

    /* ---- compressed 8-wide tree for the big-scene ray kernels ---- */
    buildWide(out, root2, rootBox, children2);
    std::vector<float>().swap(out.nodes2);
    out.buildMs = (float) std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

} // namespace pt
