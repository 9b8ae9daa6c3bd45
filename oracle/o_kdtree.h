/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h).
 *
 * o_kdtree.h: SAH kd-tree over triangles, restating (file:line under /root/reference):
 *   include/mitsuba/render/gkdtree.h:452-601     KDNode, 8-byte inner/leaf encoding
 *   include/mitsuba/render/gkdtree.h:732-745,958-1263  build parameters, scene-box enlargement
 *   include/mitsuba/render/gkdtree.h:1792-1925   buildTreeMinMax (min-max binning above 65536 prims)
 *   include/mitsuba/render/gkdtree.h:1954-2400   buildTree (O(n log n) sweep, perfect splits, retraction)
 *   include/mitsuba/render/sahkdtree3.h:39-84    SurfaceAreaHeuristic3
 *   include/mitsuba/render/sahkdtree3.h:138-308  HashedMailbox + rayIntersectHavran (TA^B_rec)
 *   include/mitsuba/render/triaccel.h:37-158     TriAccel::load / rayIntersect (Wald projection)
 *   src/libcore/triangle.cpp:71-147              Sutherland-Hodgman clipped AABB (double precision)
 *   include/mitsuba/core/aabb.h:308-338          AABB::rayIntersect slab test
 *   src/librender/skdtree.cpp:112-142,207-226    ShapeKDTree::rayIntersect (closest / shadow)
 *
 * The construction is a restatement of the same greedy SAH algorithm (same costs, same
 * termination rules, same event ordering); memory management (chunk allocators, parallel
 * subtree build, indirection nodes) is not reproduced since it does not affect query results.
 */
#pragma once
#include "o_math.h"
#include <vector>
#include <cassert>

namespace orc {

struct AABB {
    Vec3 min, max;
    AABB() { reset(); }
    AABB(const Vec3 &mi, const Vec3 &ma) : min(mi), max(ma) {}
    void reset() {
        min = Vec3(std::numeric_limits<Float>::infinity());
        max = Vec3(-std::numeric_limits<Float>::infinity());
    }
    void expandBy(const Vec3 &p) {
        for (int i = 0; i < 3; ++i) { min[i] = std::min(min[i], p[i]); max[i] = std::max(max[i], p[i]); }
    }
    void expandBy(const AABB &b) {
        for (int i = 0; i < 3; ++i) { min[i] = std::min(min[i], b.min[i]); max[i] = std::max(max[i], b.max[i]); }
    }
    void clip(const AABB &b) {
        for (int i = 0; i < 3; ++i) { min[i] = std::max(min[i], b.min[i]); max[i] = std::min(max[i], b.max[i]); }
    }
    bool isValid() const { for (int i = 0; i < 3; ++i) if (max[i] < min[i]) return false; return true; }
    Vec3 getExtents() const { return max - min; }
    Float getSurfaceArea() const { Vec3 d = max - min; return (Float) 2.0 * (d.x * d.y + d.x * d.z + d.y * d.z); }

    /* aabb.h:308-338 */
    bool rayIntersect(const Vec3 &o, const Vec3 &d, const Vec3 &dRcp, Float &nearT, Float &farT) const {
        nearT = -std::numeric_limits<Float>::infinity();
        farT = std::numeric_limits<Float>::infinity();
        for (int i = 0; i < 3; i++) {
            const Float origin = o[i];
            const Float minVal = min[i], maxVal = max[i];
            if (d[i] == 0) {
                if (origin < minVal || origin > maxVal)
                    return false;
            } else {
                Float t1 = (minVal - origin) * dRcp[i];
                Float t2 = (maxVal - origin) * dRcp[i];
                if (t1 > t2) std::swap(t1, t2);
                nearT = std::max(t1, nearT);
                farT = std::min(t2, farT);
                if (!(nearT <= farT))
                    return false;
            }
        }
        return true;
    }
};

struct Ray {
    Vec3 o; Float mint; Vec3 d; Float maxt; Vec3 dRcp;
    Ray() : mint(ORC_EPSILON), maxt(std::numeric_limits<Float>::infinity()) {}
    Ray(const Vec3 &o_, const Vec3 &d_) : o(o_), mint(ORC_EPSILON), d(d_), maxt(std::numeric_limits<Float>::infinity()) { setDir(); }
    Ray(const Vec3 &o_, const Vec3 &d_, Float mint_, Float maxt_) : o(o_), mint(mint_), d(d_), maxt(maxt_) { setDir(); }
    void setDir() { for (int i = 0; i < 3; ++i) dRcp[i] = (Float) 1 / d[i]; }   /* ray.h:77-85 */
    Vec3 operator()(Float t) const { return o + d * t; }
};

/* triaccel.h:37-158 */
struct TriAccel {
    uint32_t k;
    Float n_u, n_v, n_d;
    Float a_u, a_v, b_nu, b_nv;
    Float c_nu, c_nv;
    uint32_t shapeIndex, primIndex;

    int load(const Vec3 &A, const Vec3 &B, const Vec3 &C) {
        static const int waldModulo[4] = { 1, 2, 0, 1 };
        Vec3 b = C - A, c = B - A, N = cross(c, b);
        k = 0;
        for (int j = 0; j < 3; j++)
            if (std::abs(N[j]) > std::abs(N[k]))
                k = j;
        uint32_t u = waldModulo[k], v = waldModulo[k + 1];
        const Float n_k = N[k], denom = b[u] * c[v] - b[v] * c[u];
        if (denom == 0) {
            k = 3;
            n_u = n_v = n_d = a_u = a_v = b_nu = b_nv = c_nu = c_nv = 0;
            return 1;
        }
        n_u = N[u] / n_k;
        n_v = N[v] / n_k;
        n_d = dot(A, N) / n_k;
        b_nu = b[u] / denom;
        b_nv = -b[v] / denom;
        a_u = A[u];
        a_v = A[v];
        c_nu = c[v] / denom;
        c_nv = -c[u] / denom;
        return 0;
    }

    inline bool rayIntersect(const Ray &ray, Float mint, Float maxt, Float &u, Float &v, Float &t) const {
        Float o_u, o_v, o_k, d_u, d_v, d_k;
        switch (k) {
            case 0: o_u = ray.o[1]; o_v = ray.o[2]; o_k = ray.o[0]; d_u = ray.d[1]; d_v = ray.d[2]; d_k = ray.d[0]; break;
            case 1: o_u = ray.o[2]; o_v = ray.o[0]; o_k = ray.o[1]; d_u = ray.d[2]; d_v = ray.d[0]; d_k = ray.d[1]; break;
            case 2: o_u = ray.o[0]; o_v = ray.o[1]; o_k = ray.o[2]; d_u = ray.d[0]; d_v = ray.d[1]; d_k = ray.d[2]; break;
            default: return false;
        }
        t = (n_d - o_u * n_u - o_v * n_v - o_k) / (d_u * n_u + d_v * n_v + d_k);
        if (t < mint || t > maxt)
            return false;
        const Float hu = o_u + t * d_u - a_u;
        const Float hv = o_v + t * d_v - a_v;
        u = hv * b_nu + hu * b_nv;
        v = hu * c_nu + hv * c_nv;
        return u >= 0 && v >= 0 && u + v <= 1.0f;
    }
};

/* ---- triangle.cpp:71-147: clipped AABB in double precision ---- */
namespace detail {
struct P3d { double v[3]; double &operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
inline float castflt_up(double val) { /* math.h:284-294 */
    float a = (float) val; int32_t b; memcpy(&b, &a, 4);
    if ((double) a < val) { b += a < 0 ? -1 : 1; memcpy(&a, &b, 4); }
    return a;
}
inline float castflt_down(double val) { /* math.h:300-310 */
    float a = (float) val; int32_t b; memcpy(&b, &a, 4);
    if ((double) a > val) { b += a > 0 ? -1 : 1; memcpy(&a, &b, 4); }
    return a;
}
inline int sutherlandHodgman(const P3d *input, int inCount, P3d *output, int axis, double splitPos, bool isMinimum) {
    if (inCount < 3) return 0;
    P3d cur = input[0];
    double sign = isMinimum ? 1.0f : -1.0f;
    double distance = sign * (cur[axis] - splitPos);
    bool curIsInside = (distance >= 0);
    int outCount = 0;
    for (int i = 0; i < inCount; ++i) {
        int nextIdx = i + 1;
        if (nextIdx == inCount) nextIdx = 0;
        P3d next = input[nextIdx];
        distance = sign * (next[axis] - splitPos);
        bool nextIsInside = (distance >= 0);
        if (curIsInside && nextIsInside) {
            output[outCount++] = next;
        } else if (curIsInside && !nextIsInside) {
            double t = (splitPos - cur[axis]) / (next[axis] - cur[axis]);
            P3d p; for (int j = 0; j < 3; ++j) p[j] = cur[j] + (next[j] - cur[j]) * t;
            p[axis] = splitPos;
            output[outCount++] = p;
        } else if (!curIsInside && nextIsInside) {
            double t = (splitPos - cur[axis]) / (next[axis] - cur[axis]);
            P3d p; for (int j = 0; j < 3; ++j) p[j] = cur[j] + (next[j] - cur[j]) * t;
            p[axis] = splitPos;
            output[outCount++] = p;
            output[outCount++] = next;
        }
        cur = next;
        curIsInside = nextIsInside;
    }
    return outCount;
}
}

inline AABB triangleClippedAABB(const Vec3 &p0, const Vec3 &p1, const Vec3 &p2, const AABB &aabb) {
    detail::P3d v1[10], v2[10];
    int n = 3;
    const Vec3 *ps[3] = { &p0, &p1, &p2 };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v1[i][j] = (double) (*ps[i])[j];
    for (int axis = 0; axis < 3; ++axis) {
        n = detail::sutherlandHodgman(v1, n, v2, axis, aabb.min[axis], true);
        n = detail::sutherlandHodgman(v2, n, v1, axis, aabb.max[axis], false);
    }
    AABB result;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 3; ++j) {
            double pos = v1[i][j];
            result.min[j] = std::min(result.min[j], detail::castflt_down(pos));
            result.max[j] = std::max(result.max[j], detail::castflt_up(pos));
        }
    result.clip(aabb);
    return result;
}

/* gkdtree.h:452-601 */
struct KDNode {
    uint32_t combined;
    union { float split; uint32_t end; };
    void initLeaf(uint32_t offset, uint32_t numPrims) { combined = 0x80000000u | offset; end = offset + numPrims; }
    void initInner(int axis, float s, uint32_t relOffset) { combined = (uint32_t) axis | (relOffset << 2); split = s; }
    bool isLeaf() const { return combined & 0x80000000u; }
    uint32_t primStart() const { return combined & 0x7fffffffu; }
    uint32_t primEnd() const { return end; }
    int axis() const { return combined & 0x3; }
    uint32_t leftOffset() const { return (combined & ~(0x3u + 0x40000000u)) >> 2; }
};
static_assert(sizeof(KDNode) == 8, "KDNode must be 8 bytes");

struct TraversalCounters {
    uint64_t nodeVisits = 0, triTests = 0, leafVisits = 0;
};

class KDTree {
public:
    /* build parameters, gkdtree.h:732-745 */
    Float traversalCost = 15, queryCost = 20, emptySpaceBonus = 0.9f;
    uint32_t stopPrims = 6, maxBadRefines = 3, exactPrimThreshold = 65536, minMaxBinCount = 128;
    uint32_t maxDepth = 0;
    bool clip = true, retract = true;
    /* test hook (oracle_scene_set_bruteforce): answer ray queries by testing EVERY TriAccel inside the clipped ray interval -- the
       structure-independent answer.  The reference's kd-tree loses a hit that lies exactly on a triangle's silhouette edge when
       that edge is a split plane and the Wald distance comes out an ulp beyond the leaf's exit distance (sahkdtree3.h:263-272
       tests against [searchStart, searchEnd] of the leaf); a BVH finds it.  Rate: ~2e-8 per sample on the Cornell box. */
    bool bruteForce = false;
    template <bool shadowRay> bool sweep(const Ray &ray, Float mint, Float maxt, Float &t, Float &uOut, Float &vOut, uint32_t &primOut) const {
        bool found = false;
        for (uint32_t p = 0; p < primCount; ++p) {
            Float u, v, tt;
            if (triAccel[p].rayIntersect(ray, mint, maxt, u, v, tt)) {
                if (shadowRay) return true;
                maxt = tt; t = tt; uOut = u; vOut = v; primOut = p; found = true;
            }
        }
        return found;
    }

    std::vector<KDNode> nodes;
    std::vector<uint32_t> indices;
    std::vector<TriAccel> triAccel;
    AABB aabb, tightAABB;
    /* statistics (gkdtree.h:1252-1256) */
    double expTraversalSteps = 0, expLeavesVisited = 0, expPrimitivesIntersected = 0, sahCost = 0;
    uint32_t retractedSplits = 0, pruned = 0, builtDepth = 0;

    const float *positions = nullptr;   /* 3*nV */
    const uint32_t *tris = nullptr;     /* 3*nT */
    uint32_t primCount = 0;

    Vec3 P(uint32_t tri, int c) const { const float *p = positions + 3 * (size_t) tris[3 * (size_t) tri + c]; return Vec3(p[0], p[1], p[2]); }

    AABB primAABB(uint32_t i) const { AABB b; b.expandBy(P(i, 0)); b.expandBy(P(i, 1)); b.expandBy(P(i, 2)); return b; }
    AABB primClippedAABB(uint32_t i, const AABB &box) const { return triangleClippedAABB(P(i, 0), P(i, 1), P(i, 2), box); }

    void build(const float *pos, const uint32_t *tri, uint32_t nTri, const uint32_t *shapeOfTri, const uint32_t *primInShape);

    /* skdtree.cpp:112-142 (closest) -- returns prim id through `prim`, barycentrics u,v */
    bool rayIntersect(const Ray &ray, Float &t, Float &u, Float &v, uint32_t &prim, TraversalCounters *ctr) const {
        Float mint, maxt;
        t = std::numeric_limits<Float>::infinity();
        if (aabb.rayIntersect(ray.o, ray.d, ray.dRcp, mint, maxt)) {
            Float rayMinT = ray.mint;
            if (rayMinT == ORC_EPSILON)
                rayMinT *= std::max(std::max(std::max(std::abs(ray.o.x), std::abs(ray.o.y)), std::abs(ray.o.z)), ORC_EPSILON);
            if (rayMinT > mint) mint = rayMinT;
            if (ray.maxt < maxt) maxt = ray.maxt;
            if (maxt > mint)
                return bruteForce ? sweep<false>(ray, mint, maxt, t, u, v, prim) : havran<false>(ray, mint, maxt, t, u, v, prim, ctr);
        }
        return false;
    }

    /* skdtree.cpp:207-226 (shadow) */
    bool rayIntersectShadow(const Ray &ray, TraversalCounters *ctr) const {
        Float mint, maxt, t = std::numeric_limits<Float>::infinity(), u, v; uint32_t prim;
        if (aabb.rayIntersect(ray.o, ray.d, ray.dRcp, mint, maxt)) {
            Float rayMinT = ray.mint;
            if (rayMinT == ORC_EPSILON)
                rayMinT *= std::max(std::max(std::abs(ray.o.x), std::abs(ray.o.y)), std::abs(ray.o.z));
            if (rayMinT > mint) mint = rayMinT;
            if (ray.maxt < maxt) maxt = ray.maxt;
            if (maxt > mint)
                if (bruteForce ? sweep<true>(ray, mint, maxt, t, u, v, prim) : havran<true>(ray, mint, maxt, t, u, v, prim, ctr))
                    return true;
        }
        return false;
    }

private:
    struct StackEntry { const KDNode *node; Float t; uint32_t prev; Vec3 p; };

    /* sahkdtree3.h:178-308 */
    template <bool shadowRay>
    bool havran(const Ray &ray, Float mint, Float maxt, Float &t, Float &uOut, Float &vOut, uint32_t &primOut, TraversalCounters *ctr) const {
        StackEntry stack[48];
        uint32_t mailbox[8];
        memset(mailbox, 0xFF, sizeof(mailbox));

        uint32_t enPt = 0;
        stack[enPt].t = mint;
        stack[enPt].p = ray(mint);
        uint32_t exPt = 1;
        stack[exPt].t = maxt;
        stack[exPt].p = ray(maxt);
        stack[exPt].node = nullptr;

        bool foundIntersection = false;
        const KDNode *currNode = nodes.data();
        while (currNode != nullptr) {
            while (!currNode->isLeaf()) {
                if (ctr) ctr->nodeVisits++;
                const Float splitVal = currNode->split;
                const int axis = currNode->axis();
                const KDNode *farChild;
                const KDNode *left = currNode + currNode->leftOffset();
                if (stack[enPt].p[axis] <= splitVal) {
                    if (stack[exPt].p[axis] <= splitVal) { currNode = left; continue; }
                    if (stack[enPt].p[axis] == splitVal) { currNode = left + 1; continue; }
                    currNode = left;
                    farChild = currNode + 1;
                } else {
                    if (splitVal < stack[exPt].p[axis]) { currNode = left + 1; continue; }
                    farChild = left;
                    currNode = farChild + 1;
                }
                Float distToSplit = (splitVal - ray.o[axis]) * ray.dRcp[axis];
                const uint32_t tmp = exPt++;
                if (exPt == enPt) ++exPt;
                assert(exPt < 48);
                stack[exPt].prev = tmp;
                stack[exPt].t = distToSplit;
                stack[exPt].node = farChild;
                stack[exPt].p = ray(distToSplit);
                stack[exPt].p[axis] = splitVal;
            }
            if (ctr) ctr->leafVisits++;
            for (uint32_t entry = currNode->primStart(), last = currNode->primEnd(); entry != last; entry++) {
                const uint32_t primIdx = indices[entry];
                if (mailbox[primIdx & 7] == primIdx)
                    continue;
                if (ctr) ctr->triTests++;
                Float tu, tv, tt;
                bool result = triAccel[primIdx].rayIntersect(ray, mint, maxt, tu, tv, tt);
                if (result) {
                    if (shadowRay) return true;
                    maxt = tt; t = tt; uOut = tu; vOut = tv; primOut = primIdx;
                    foundIntersection = true;
                }
                mailbox[primIdx & 7] = primIdx;
            }
            if (stack[exPt].t > maxt)
                break;
            enPt = exPt;
            currNode = stack[exPt].node;
            exPt = stack[enPt].prev;
        }
        return foundIntersection;
    }

    /* ---------------- construction ---------------- */
    enum { EEdgeEnd = 0, EEdgePlanar = 1, EEdgeStart = 2 };
    struct EdgeEvent {
        float pos; uint32_t index; uint16_t type; uint16_t axis;
        EdgeEvent() {}
        EdgeEvent(int type_, int axis_, float pos_, uint32_t idx) : pos(pos_), index(idx), type((uint16_t) type_), axis((uint16_t) axis_) {}
    };
    struct EdgeEventOrdering { /* gkdtree.h:1331-1341 */
        bool operator()(const EdgeEvent &a, const EdgeEvent &b) const {
            if (a.axis != b.axis) return a.axis < b.axis;
            if (a.pos != b.pos) return a.pos < b.pos;
            if (a.type != b.type) return a.type < b.type;
            return a.index < b.index;
        }
    };
    struct SAH { /* sahkdtree3.h:39-84 */
        Vec3 temp0, temp1;
        explicit SAH(const AABB &b) {
            const Vec3 e = b.getExtents();
            const Float temp = 1.0f / (e.x * e.y + e.y * e.z + e.x * e.z);
            temp0 = Vec3(e[1] * e[2], e[0] * e[2], e[0] * e[1]) * temp;
            temp1 = Vec3(e[1] + e[2], e[0] + e[2], e[0] + e[1]) * temp;
        }
        std::pair<Float, Float> operator()(int axis, Float l, Float r) const {
            return std::make_pair(temp0[axis] + temp1[axis] * l, temp0[axis] + temp1[axis] * r);
        }
    };
    struct SplitCandidate {
        Float cost = std::numeric_limits<Float>::infinity();
        float pos = 0; int axis = 0; uint32_t numLeft = 0, numRight = 0; bool planarLeft = false;
    };

    std::vector<uint8_t> classStorage;
    enum { EBothSides = 1, ELeftSide = 2, ERightSide = 3, EBothSidesProcessed = 4 };

    void createLeafFromEvents(uint32_t nodeIdx, const std::vector<EdgeEvent> &ev, uint32_t primCount_) {
        uint32_t start = (uint32_t) indices.size();
        uint32_t seen = 0;
        for (size_t i = 0; i < ev.size() && ev[i].axis == 0; ++i)
            if (ev[i].type == EEdgeStart || ev[i].type == EEdgePlanar) { indices.push_back(ev[i].index); seen++; }
        assert(seen == primCount_);
        nodes[nodeIdx].initLeaf(start, primCount_);
    }
    void createLeafFromIndices(uint32_t nodeIdx, const std::vector<uint32_t> &idx) {
        uint32_t start = (uint32_t) indices.size();
        indices.insert(indices.end(), idx.begin(), idx.end());
        nodes[nodeIdx].initLeaf(start, (uint32_t) idx.size());
    }
    void createLeafAfterRetraction(uint32_t nodeIdx, uint32_t start) { /* gkdtree.h:1665-1700 */
        std::vector<uint32_t> tmp(indices.begin() + start, indices.end());
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        indices.resize(start);
        indices.insert(indices.end(), tmp.begin(), tmp.end());
        nodes[nodeIdx].initLeaf(start, (uint32_t) tmp.size());
    }

    Float buildMinMax(uint32_t depth, uint32_t nodeIdx, const AABB &nodeAABB, const AABB &tight,
                      std::vector<uint32_t> &prims, uint32_t badRefines);
    Float transitionToNLogN(uint32_t depth, uint32_t nodeIdx, const AABB &nodeAABB,
                            std::vector<uint32_t> &prims, uint32_t badRefines);
    Float buildExact(uint32_t depth, uint32_t nodeIdx, const AABB &nodeAABB,
                     std::vector<EdgeEvent> &events, uint32_t primCount_, uint32_t badRefines);
    void computeStatistics();
};

/* ====================================================================== */

inline void KDTree::build(const float *pos, const uint32_t *tri, uint32_t nTri, const uint32_t *shapeOfTri, const uint32_t *primInShape) {
    positions = pos; tris = tri; primCount = nTri;
    nodes.clear(); indices.clear();
    /* TriAccel precompute, skdtree.cpp:76-110 */
    triAccel.resize(nTri);
    for (uint32_t i = 0; i < nTri; ++i) {
        triAccel[i].load(P(i, 0), P(i, 1), P(i, 2));
        triAccel[i].shapeIndex = shapeOfTri ? shapeOfTri[i] : 0;
        triAccel[i].primIndex = primInShape ? primInShape[i] : i;
    }
    if (nTri == 0) {
        nodes.resize(1); nodes[0].initLeaf(0, 0);
        aabb.reset(); tightAABB = aabb;
        return;
    }
    if (maxDepth == 0) {
        int log2i = 0; { uint32_t v = nTri; while (v >>= 1) ++log2i; }   /* math::log2i */
        maxDepth = (uint32_t) (8 + 1.3f * log2i);
    }
    maxDepth = std::min(maxDepth, 48u);

    AABB box;
    std::vector<uint32_t> prims(nTri);
    for (uint32_t i = 0; i < nTri; ++i) { box.expandBy(primAABB(i)); prims[i] = i; }
    classStorage.assign(nTri, 0);

    nodes.reserve(nTri);
    nodes.resize(1);
    buildMinMax(1, 0, box, box, prims, 0);

    /* gkdtree.h:1213-1220 */
    tightAABB = box;
    aabb = box;
    const Float eps = 1e-3f;
    aabb.min -= (aabb.max - aabb.min) * eps + Vec3(eps);
    aabb.max += (aabb.max - aabb.min) * eps + Vec3(eps);
    computeStatistics();
}

inline Float KDTree::transitionToNLogN(uint32_t depth, uint32_t nodeIdx, const AABB &nodeAABB,
                                       std::vector<uint32_t> &prims, uint32_t badRefines) {
    /* gkdtree.h:1702-1790: create the initial edge-event list, clipping to the node box */
    std::vector<EdgeEvent> events;
    events.reserve(prims.size() * 6);
    uint32_t actualPrimCount = 0;
    for (uint32_t index : prims) {
        AABB b;
        if (clip) {
            b = primClippedAABB(index, nodeAABB);
            if (!b.isValid() || b.getSurfaceArea() == 0)
                continue;
        } else {
            b = primAABB(index);
        }
        for (int axis = 0; axis < 3; ++axis) {
            float mn = b.min[axis], mx = b.max[axis];
            if (mn == mx) {
                events.push_back(EdgeEvent(EEdgePlanar, axis, mn, index));
            } else {
                events.push_back(EdgeEvent(EEdgeStart, axis, mn, index));
                events.push_back(EdgeEvent(EEdgeEnd, axis, mx, index));
            }
        }
        ++actualPrimCount;
    }
    std::sort(events.begin(), events.end(), EdgeEventOrdering());
    std::vector<uint32_t>().swap(prims);
    return buildExact(depth, nodeIdx, nodeAABB, events, actualPrimCount, badRefines);
}

inline Float KDTree::buildMinMax(uint32_t depth, uint32_t nodeIdx, const AABB &nodeAABB, const AABB &tight,
                                 std::vector<uint32_t> &prims, uint32_t badRefines) {
    const uint32_t primCount_ = (uint32_t) prims.size();
    builtDepth = std::max(builtDepth, depth);
    Float leafCost = primCount_ * queryCost;
    if (primCount_ <= stopPrims || depth >= maxDepth) {
        createLeafFromIndices(nodeIdx, prims);
        return leafCost;
    }
    if (primCount_ <= exactPrimThreshold)
        return transitionToNLogN(depth, nodeIdx, nodeAABB, prims, badRefines);

    /* ---- min-max binning, gkdtree.h:2405-2630 ---- */
    const uint32_t B = minMaxBinCount;
    std::vector<uint32_t> minBins(3 * B, 0), maxBins(3 * B, 0);
    Vec3 invBinSize;
    for (int a = 0; a < 3; ++a) invBinSize[a] = 1 / ((tight.max[a] - tight.min[a]) / B);
    const int64_t maxBin = (int64_t) B - 1;
    for (uint32_t index : prims) {
        AABB b = primAABB(index);
        for (int a = 0; a < 3; ++a) {
            int64_t mi = (int64_t) ((b.min[a] - tight.min[a]) * invBinSize[a]);
            int64_t ma = (int64_t) ((b.max[a] - tight.min[a]) * invBinSize[a]);
            mi = std::max<int64_t>(0, std::min(mi, maxBin));
            ma = std::max<int64_t>(0, std::min(ma, maxBin));
            minBins[a * B + mi]++;
            maxBins[a * B + ma]++;
        }
    }
    SplitCandidate best;
    int bestBin = -1;
    {
        SAH tch(tight);
        for (int axis = 0; axis < 3; ++axis) {
            uint32_t numLeft = 0, numRight = primCount_;
            Float leftWidth = 0, rightWidth = tight.max[axis] - tight.min[axis];
            const Float binSize = rightWidth / B;
            for (uint32_t i = 0; i + 1 < B; ++i) {
                numLeft += minBins[axis * B + i];
                numRight -= maxBins[axis * B + i];
                leftWidth += binSize;
                rightWidth -= binSize;
                std::pair<Float, Float> prob = tch(axis, leftWidth, rightWidth);
                Float cost = traversalCost + queryCost * (prob.first * numLeft + prob.second * numRight);
                if (cost < best.cost) {
                    best.cost = cost; best.axis = axis; best.numLeft = numLeft; best.numRight = numRight;
                    bestBin = (int) i;
                }
            }
        }
        if (bestBin >= 0) {
            /* split plane = upper edge of the chosen bin */
            const Float binSize = (tight.max[best.axis] - tight.min[best.axis]) / B;
            best.pos = tight.min[best.axis] + binSize * (bestBin + 1);
            if (!(best.pos > nodeAABB.min[best.axis] && best.pos < nodeAABB.max[best.axis]))
                best.cost = std::numeric_limits<Float>::infinity();
        }
    }
    if (best.cost == std::numeric_limits<Float>::infinity())
        return transitionToNLogN(depth, nodeIdx, nodeAABB, prims, badRefines);

    if (best.cost >= leafCost) {
        if ((best.cost > 4 * leafCost && primCount_ < 16) || badRefines >= maxBadRefines) {
            createLeafFromIndices(nodeIdx, prims);
            return leafCost;
        }
        ++badRefines;
    }

    /* partition: a primitive goes left if its min bin <= bestBin, right if its max bin > bestBin */
    std::vector<uint32_t> leftPrims, rightPrims;
    AABB leftTight, rightTight;
    {
        const int a = best.axis;
        for (uint32_t index : prims) {
            AABB b = primAABB(index);
            int64_t mi = (int64_t) ((b.min[a] - tight.min[a]) * invBinSize[a]);
            int64_t ma = (int64_t) ((b.max[a] - tight.min[a]) * invBinSize[a]);
            mi = std::max<int64_t>(0, std::min(mi, maxBin));
            ma = std::max<int64_t>(0, std::min(ma, maxBin));
            if (mi <= bestBin) {
                leftPrims.push_back(index);
                AABB c = b; c.clip(nodeAABB); c.max[a] = std::min(c.max[a], best.pos); leftTight.expandBy(c);
            }
            if (ma > bestBin) {
                rightPrims.push_back(index);
                AABB c = b; c.clip(nodeAABB); c.min[a] = std::max(c.min[a], best.pos); rightTight.expandBy(c);
            }
        }
    }
    std::vector<uint32_t>().swap(prims);

    uint32_t children = (uint32_t) nodes.size();
    nodes.resize(nodes.size() + 2);
    const uint32_t nodePosBeforeSplit = (uint32_t) nodes.size();
    const uint32_t indexPosBeforeSplit = (uint32_t) indices.size();
    nodes[nodeIdx].initInner(best.axis, best.pos, children - nodeIdx);

    AABB childAABB(nodeAABB);
    childAABB.max[best.axis] = best.pos;
    AABB lt = leftTight; lt.clip(childAABB);
    Float leftCost = buildMinMax(depth + 1, children, childAABB, lt, leftPrims, badRefines);
    childAABB.min[best.axis] = best.pos;
    childAABB.max[best.axis] = nodeAABB.max[best.axis];
    AABB rt = rightTight; rt.clip(childAABB);
    Float rightCost = buildMinMax(depth + 1, children + 1, childAABB, rt, rightPrims, badRefines);

    SAH tch(nodeAABB);
    std::pair<Float, Float> prob = tch(best.axis, best.pos - nodeAABB.min[best.axis], nodeAABB.max[best.axis] - best.pos);
    Float finalCost = traversalCost + (prob.first * leftCost + prob.second * rightCost);
    if (!retract || finalCost < primCount_ * queryCost)
        return finalCost;
    nodes.resize(nodePosBeforeSplit);
    retractedSplits++;
    createLeafAfterRetraction(nodeIdx, indexPosBeforeSplit);
    return leafCost;
}

inline Float KDTree::buildExact(uint32_t depth, uint32_t nodeIdx, const AABB &nodeAABB,
                                std::vector<EdgeEvent> &events, uint32_t primCount_, uint32_t badRefines) {
    builtDepth = std::max(builtDepth, depth);
    Float leafCost = primCount_ * queryCost;
    if (primCount_ <= stopPrims || depth >= maxDepth) {
        createLeafFromEvents(nodeIdx, events, primCount_);
        return leafCost;
    }
    SplitCandidate best;

    /* ---- split candidate search, gkdtree.h:1966-2090 ---- */
    uint32_t numLeft[3] = { 0, 0, 0 }, numRight[3] = { primCount_, primCount_, primCount_ };
    size_t eventsByAxis[3] = { 0, events.size(), events.size() };
    int eventsByAxisCtr = 1;
    SAH tch(nodeAABB);
    const size_t nEv = events.size();
    for (size_t e = 0; e < nEv;) {
        int axis = events[e].axis;
        float pos = events[e].pos;
        uint32_t numStart = 0, numEnd = 0, numPlanar = 0;
        while (e < nEv && events[e].pos == pos && events[e].axis == axis && events[e].type == EEdgeEnd) { ++numEnd; ++e; }
        while (e < nEv && events[e].pos == pos && events[e].axis == axis && events[e].type == EEdgePlanar) { ++numPlanar; ++e; }
        while (e < nEv && events[e].pos == pos && events[e].axis == axis && events[e].type == EEdgeStart) { ++numStart; ++e; }
        if (e < nEv && events[e].axis != axis)
            eventsByAxis[eventsByAxisCtr++] = e;
        numRight[axis] -= numPlanar + numEnd;
        if (pos > nodeAABB.min[axis] && pos < nodeAABB.max[axis]) {
            const uint32_t nL = numLeft[axis], nR = numRight[axis];
            const Float nLF = (Float) nL, nRF = (Float) nR;
            std::pair<Float, Float> prob = tch(axis, pos - nodeAABB.min[axis], nodeAABB.max[axis] - pos);
            if (numPlanar == 0) {
                Float cost = traversalCost + queryCost * (prob.first * nLF + prob.second * nRF);
                if (nL == 0 || nR == 0) cost *= emptySpaceBonus;
                if (cost < best.cost) { best.pos = pos; best.axis = axis; best.cost = cost; best.numLeft = nL; best.numRight = nR; }
            } else {
                Float costPlanarLeft = traversalCost + queryCost * (prob.first * (nL + numPlanar) + prob.second * nRF);
                Float costPlanarRight = traversalCost + queryCost * (prob.first * nLF + prob.second * (nR + numPlanar));
                if (nL + numPlanar == 0 || nR == 0) costPlanarLeft *= emptySpaceBonus;
                if (nL == 0 || nR + numPlanar == 0) costPlanarRight *= emptySpaceBonus;
                if (costPlanarLeft < best.cost || costPlanarRight < best.cost) {
                    best.pos = pos; best.axis = axis;
                    if (costPlanarLeft < costPlanarRight) {
                        best.cost = costPlanarLeft; best.numLeft = nL + numPlanar; best.numRight = nR; best.planarLeft = true;
                    } else {
                        best.cost = costPlanarRight; best.numLeft = nL; best.numRight = nR + numPlanar; best.planarLeft = false;
                    }
                }
            }
        }
        numLeft[axis] += numStart + numPlanar;
    }

    /* "bad refines" heuristic, gkdtree.h:2100-2108 */
    if (best.cost >= leafCost) {
        if ((best.cost > 4 * leafCost && primCount_ < 16) || badRefines >= maxBadRefines
            || best.cost == std::numeric_limits<Float>::infinity()) {
            createLeafFromEvents(nodeIdx, events, primCount_);
            return leafCost;
        }
        ++badRefines;
    }

    /* ---- classification, gkdtree.h:2114-2166 ---- */
    const size_t axBegin = eventsByAxis[best.axis];
    for (size_t e = axBegin; e < nEv && events[e].axis == best.axis; ++e)
        classStorage[events[e].index] = EBothSides;
    uint32_t primsLeft = 0, primsRight = 0, primsBoth = primCount_;
    for (size_t e = axBegin; e < nEv && events[e].axis == best.axis; ++e) {
        const EdgeEvent &ev = events[e];
        if (ev.type == EEdgeEnd && ev.pos <= best.pos) {
            classStorage[ev.index] = ELeftSide; primsBoth--; primsLeft++;
        } else if (ev.type == EEdgeStart && ev.pos >= best.pos) {
            classStorage[ev.index] = ERightSide; primsBoth--; primsRight++;
        } else if (ev.type == EEdgePlanar) {
            if (ev.pos < best.pos || (ev.pos == best.pos && best.planarLeft)) {
                classStorage[ev.index] = ELeftSide; primsBoth--; primsLeft++;
            } else {
                classStorage[ev.index] = ERightSide; primsBoth--; primsRight++;
            }
        }
    }
    assert(primsLeft + primsBoth == best.numLeft && primsRight + primsBoth == best.numRight);

    AABB leftNodeAABB = nodeAABB, rightNodeAABB = nodeAABB;
    leftNodeAABB.max[best.axis] = best.pos;
    rightNodeAABB.min[best.axis] = best.pos;
    uint32_t prunedLeft = 0, prunedRight = 0;

    /* ---- partitioning, gkdtree.h:2190-2290 ---- */
    std::vector<EdgeEvent> leftEvents, rightEvents;
    if (clip) {
        std::vector<EdgeEvent> leftTemp, rightTemp, newLeft, newRight;
        leftTemp.reserve(primsLeft * 6); rightTemp.reserve(primsRight * 6);
        newLeft.reserve(primsBoth * 6); newRight.reserve(primsBoth * 6);
        for (size_t e = 0; e < nEv; ++e) {
            const EdgeEvent &ev = events[e];
            int cls = classStorage[ev.index];
            if (cls == ELeftSide) leftTemp.push_back(ev);
            else if (cls == ERightSide) rightTemp.push_back(ev);
            else if (cls == EBothSides) {
                const uint32_t index = ev.index;
                AABB cl = primClippedAABB(index, leftNodeAABB);
                AABB cr = primClippedAABB(index, rightNodeAABB);
                if (cl.isValid() && cl.getSurfaceArea() > 0) {
                    for (int axis = 0; axis < 3; ++axis) {
                        float mn = cl.min[axis], mx = cl.max[axis];
                        if (mn == mx) newLeft.push_back(EdgeEvent(EEdgePlanar, axis, mn, index));
                        else { newLeft.push_back(EdgeEvent(EEdgeStart, axis, mn, index)); newLeft.push_back(EdgeEvent(EEdgeEnd, axis, mx, index)); }
                    }
                } else prunedLeft++;
                if (cr.isValid() && cr.getSurfaceArea() > 0) {
                    for (int axis = 0; axis < 3; ++axis) {
                        float mn = cr.min[axis], mx = cr.max[axis];
                        if (mn == mx) newRight.push_back(EdgeEvent(EEdgePlanar, axis, mn, index));
                        else { newRight.push_back(EdgeEvent(EEdgeStart, axis, mn, index)); newRight.push_back(EdgeEvent(EEdgeEnd, axis, mx, index)); }
                    }
                } else prunedRight++;
                classStorage[index] = EBothSidesProcessed;
            }
        }
        pruned += prunedLeft + prunedRight;
        std::sort(newLeft.begin(), newLeft.end(), EdgeEventOrdering());
        std::sort(newRight.begin(), newRight.end(), EdgeEventOrdering());
        leftEvents.resize(leftTemp.size() + newLeft.size());
        std::merge(leftTemp.begin(), leftTemp.end(), newLeft.begin(), newLeft.end(), leftEvents.begin(), EdgeEventOrdering());
        rightEvents.resize(rightTemp.size() + newRight.size());
        std::merge(rightTemp.begin(), rightTemp.end(), newRight.begin(), newRight.end(), rightEvents.begin(), EdgeEventOrdering());
    } else {
        for (size_t e = 0; e < nEv; ++e) {
            const EdgeEvent &ev = events[e];
            int cls = classStorage[ev.index];
            if (cls == ELeftSide) leftEvents.push_back(ev);
            else if (cls == ERightSide) rightEvents.push_back(ev);
            else { leftEvents.push_back(ev); rightEvents.push_back(ev); }
        }
    }
    std::vector<EdgeEvent>().swap(events);

    /* ---- recursion, gkdtree.h:2306-2365 ---- */
    uint32_t children = (uint32_t) nodes.size();
    nodes.resize(nodes.size() + 2);
    const uint32_t nodePosBeforeSplit = (uint32_t) nodes.size();
    const uint32_t indexPosBeforeSplit = (uint32_t) indices.size();
    nodes[nodeIdx].initInner(best.axis, best.pos, children - nodeIdx);

    Float leftCost = buildExact(depth + 1, children, leftNodeAABB, leftEvents, best.numLeft - prunedLeft, badRefines);
    Float rightCost = buildExact(depth + 1, children + 1, rightNodeAABB, rightEvents, best.numRight - prunedRight, badRefines);

    std::pair<Float, Float> prob = tch(best.axis, best.pos - nodeAABB.min[best.axis], nodeAABB.max[best.axis] - best.pos);
    Float finalCost = traversalCost + (prob.first * leftCost + prob.second * rightCost);
    if (!retract || finalCost < primCount_ * queryCost)
        return finalCost;
    nodes.resize(nodePosBeforeSplit);
    retractedSplits++;
    createLeafAfterRetraction(nodeIdx, indexPosBeforeSplit);
    return leafCost;
}

inline void KDTree::computeStatistics() {
    /* gkdtree.h:1100-1211: SAH expectations per random query */
    struct Item { uint32_t node; AABB box; };
    std::vector<Item> stack;
    stack.push_back({ 0, tightAABB });
    expTraversalSteps = expLeavesVisited = expPrimitivesIntersected = sahCost = 0;
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        const KDNode &n = nodes[it.node];
        Float q = it.box.getSurfaceArea();
        if (n.isLeaf()) {
            uint32_t c = n.primEnd() - n.primStart();
            expLeavesVisited += q; expPrimitivesIntersected += (double) q * c; sahCost += (double) q * c * queryCost;
        } else {
            expTraversalSteps += q; sahCost += q * traversalCost;
            uint32_t l = it.node + n.leftOffset();
            AABB lb = it.box, rb = it.box;
            lb.max[n.axis()] = n.split; rb.min[n.axis()] = n.split;
            stack.push_back({ l, lb }); stack.push_back({ l + 1, rb });
        }
    }
    Float root = tightAABB.getSurfaceArea();
    if (root > 0) { expTraversalSteps /= root; expLeavesVisited /= root; expPrimitivesIntersected /= root; sahCost /= root; }
}

} // namespace orc
