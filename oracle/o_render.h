/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h).
 *
 * o_render.h: camera, reconstruction filter, image blocks and the block-parallel render loop.
 * Restates (file:line under /root/reference):
 *   src/sensors/perspective.cpp:126-180,271-297  camera-to-sample transform, sampleRayDifferential
 *   src/libcore/transform.cpp:33-63,99-123       translate / scale / perspective (+ their inverses)
 *   include/mitsuba/core/matrix.inl:138-193      Gauss-Jordan inverse used by Transform(Matrix4x4)
 *   include/mitsuba/core/transform.h:108-183     Transform applied to Point / Vector
 *   src/librender/sensor.cpp:84-110              aspect, resolution, invResolution
 *   src/libcore/rfilter.cpp:38-57, include/mitsuba/core/rfilter.h:28,76-77, src/rfilters/gaussian.cpp:34-57
 *   include/mitsuba/render/imageblock.h:103-204  ImageBlock::put(sample) / put(block)
 *   src/librender/imageproc.cpp:28-78            spiral block order
 *   src/librender/integrator.cpp:140-188         renderBlock
 *   src/librender/renderproc.cpp:68-86,142-149   per-block processing, film merge
 *   src/libcore/fmtconv.cpp:979-991              weight normalisation
 * Per-pixel order inside a block is row-major here (the reference walks a Hilbert curve,
 * renderproc.cpp:80-82); with the ctr stream the order only permutes float additions.
 */
#pragma once
#include "o_direct.h"
#include "o_volpath.h"
#include <thread>
#include <atomic>
#include <mutex>
#include <chrono>

namespace orc {

struct Transform { /* forward + tracked inverse, like mitsuba::Transform */
    Mat4 fwd, inv;
    static Transform translate(const Vec3 &v) {
        Transform t; t.fwd = Mat4::identity(); t.inv = Mat4::identity();
        t.fwd.m[0][3] = v.x; t.fwd.m[1][3] = v.y; t.fwd.m[2][3] = v.z;
        t.inv.m[0][3] = -v.x; t.inv.m[1][3] = -v.y; t.inv.m[2][3] = -v.z;
        return t;
    }
    static Transform scale(const Vec3 &v) {
        Transform t; t.fwd = Mat4::identity(); t.inv = Mat4::identity();
        t.fwd.m[0][0] = v.x; t.fwd.m[1][1] = v.y; t.fwd.m[2][2] = v.z;
        t.inv.m[0][0] = 1.0f / v.x; t.inv.m[1][1] = 1.0f / v.y; t.inv.m[2][2] = 1.0f / v.z;
        return t;
    }
    static Transform perspective(Float fov, Float clipNear, Float clipFar) { /* transform.cpp:99-123 */
        Float recip = 1.0f / (clipFar - clipNear);
        Float cot = 1.0f / om::tan((fov / 2.0f) * (ORC_PI / 180.0f));   /* degToRad = value * (M_PI/180) */
        Transform t;
        memset(t.fwd.m, 0, sizeof(t.fwd.m));
        t.fwd.m[0][0] = cot; t.fwd.m[1][1] = cot;
        t.fwd.m[2][2] = clipFar * recip; t.fwd.m[2][3] = -clipNear * clipFar * recip;
        t.fwd.m[3][2] = 1;
        t.fwd.invert(t.inv);
        return t;
    }
    Transform operator*(const Transform &o) const { Transform t; t.fwd = fwd * o.fwd; t.inv = o.inv * inv; return t; }
    Transform inverse() const { Transform t; t.fwd = inv; t.inv = fwd; return t; }
    Vec3 point(const Vec3 &p) const { /* transform.h:108-125 */
        Float x = fwd.m[0][0] * p.x + fwd.m[0][1] * p.y + fwd.m[0][2] * p.z + fwd.m[0][3];
        Float y = fwd.m[1][0] * p.x + fwd.m[1][1] * p.y + fwd.m[1][2] * p.z + fwd.m[1][3];
        Float z = fwd.m[2][0] * p.x + fwd.m[2][1] * p.y + fwd.m[2][2] * p.z + fwd.m[2][3];
        Float w = fwd.m[3][0] * p.x + fwd.m[3][1] * p.y + fwd.m[3][2] * p.z + fwd.m[3][3];
        if (w == 1.0f) return Vec3(x, y, z);
        return Vec3(x, y, z) / w;
    }
    Vec3 pointAffine(const Vec3 &p) const { /* transform.h:128-136 */
        return Vec3(fwd.m[0][0] * p.x + fwd.m[0][1] * p.y + fwd.m[0][2] * p.z + fwd.m[0][3],
                    fwd.m[1][0] * p.x + fwd.m[1][1] * p.y + fwd.m[1][2] * p.z + fwd.m[1][3],
                    fwd.m[2][0] * p.x + fwd.m[2][1] * p.y + fwd.m[2][2] * p.z + fwd.m[2][3]);
    }
    Vec3 vector(const Vec3 &v) const { /* transform.h:175-183 */
        return Vec3(fwd.m[0][0] * v.x + fwd.m[0][1] * v.y + fwd.m[0][2] * v.z,
                    fwd.m[1][0] * v.x + fwd.m[1][1] * v.y + fwd.m[1][2] * v.z,
                    fwd.m[2][0] * v.x + fwd.m[2][1] * v.y + fwd.m[2][2] * v.z);
    }
};

struct PerspectiveCamera {
    Transform sampleToCamera, toWorld;
    Float nearClip, farClip;
    Vec2 invResolution;
    Vec3 dx, dy;                         /* position differentials on the near plane, perspective.cpp:159-163 */

    void configure(const phip_camera &c, const phip_film &f) { /* perspective.cpp:126-157, sensor.cpp:104-109 */
        Float aspect = f.width / (Float) f.height;
        invResolution = Vec2((Float) 1 / (Float) f.crop_width, (Float) 1 / (Float) f.crop_height);
        Vec2 relSize((Float) f.crop_width / (Float) f.width, (Float) f.crop_height / (Float) f.height);
        Vec2 relOffset((Float) f.crop_offset_x / (Float) f.width, (Float) f.crop_offset_y / (Float) f.height);
        nearClip = c.near_clip; farClip = c.far_clip;
        Transform cameraToSample =
              Transform::scale(Vec3(1.0f / relSize.x, 1.0f / relSize.y, 1.0f))
            * Transform::translate(Vec3(-relOffset.x, -relOffset.y, 0.0f))
            * Transform::scale(Vec3(-0.5f, -0.5f * aspect, 1.0f))
            * Transform::translate(Vec3(-1.0f, -1.0f / aspect, 0.0f))
            * Transform::perspective(c.xfov_deg, nearClip, farClip);
        sampleToCamera = cameraToSample.inverse();
        dx = sampleToCamera.point(Vec3(invResolution.x, 0.0f, 0.0f)) - sampleToCamera.point(Vec3(0.0f));
        dy = sampleToCamera.point(Vec3(0.0f, invResolution.y, 0.0f)) - sampleToCamera.point(Vec3(0.0f));
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) toWorld.fwd.m[i][j] = c.to_world[4 * i + j];
    }

    /* perspective.cpp:271-297; rx / ry receive the differential directions (rxOrigin = ryOrigin = o) */
    Ray sampleRay(const Vec2 &pixelSample, Vec3 *rx = nullptr, Vec3 *ry = nullptr) const {
        Vec3 nearP = sampleToCamera.point(Vec3(pixelSample.x * invResolution.x, pixelSample.y * invResolution.y, 0.0f));
        Vec3 d = normalize(nearP);
        Float invZ = 1.0f / d.z;
        Ray ray;
        ray.mint = nearClip * invZ;
        ray.maxt = farClip * invZ;
        ray.o = toWorld.pointAffine(Vec3(0.0f));
        ray.d = toWorld.vector(d);
        ray.setDir();
        if (rx) *rx = toWorld.vector(normalize(nearP + dx));
        if (ry) *ry = toWorld.vector(normalize(nearP + dy));
        return ray;
    }
};

/* rfilter.cpp:38-57 + gaussian.cpp:34-57: table the host side hands to phip_film.filter_table */
inline void gaussianFilterTable(Float stddev, Float &radius, Float table[PHIP_FILTER_RESOLUTION + 1]) {
    radius = 4 * stddev;
    Float sum = 0.0f;
    const Float alpha = -1.0f / (2.0f * stddev * stddev);
    for (size_t i = 0; i < PHIP_FILTER_RESOLUTION; ++i) {
        Float x = (radius * i) / PHIP_FILTER_RESOLUTION;
        Float value = std::max((Float) 0.0f, om::fastexp(alpha * x * x) - om::fastexp(alpha * radius * radius));
        table[i] = value;
        sum += value;
    }
    table[PHIP_FILTER_RESOLUTION] = 0.0f;
    sum *= 2 * radius / PHIP_FILTER_RESOLUTION;
    Float normalization = 1.0f / sum;
    for (size_t i = 0; i < PHIP_FILTER_RESOLUTION; ++i)
        table[i] *= normalization;
}

struct Filter {
    Float radius, scaleFactor; int borderSize;
    Float values[PHIP_FILTER_RESOLUTION + 1];
    void configure(const phip_film &f) {
        radius = f.filter_radius;
        memcpy(values, f.filter_table, sizeof(values));
        scaleFactor = PHIP_FILTER_RESOLUTION / radius;
        borderSize = (int) std::ceil(radius - 0.5f);
    }
    Float evalDiscretized(Float x) const { return values[std::min((int) std::abs(x * scaleFactor), PHIP_FILTER_RESOLUTION)]; }
};

/* imageblock.h: 5-channel (R,G,B,alpha,weight) float block with a border */
struct ImageBlock {
    int offX = 0, offY = 0, sizeX = 0, sizeY = 0, border = 0;
    std::vector<Float> data;
    const Filter *filter = nullptr;
    std::vector<Float> weightsX, weightsY;

    void init(int sx, int sy, const Filter *f, bool withBorder) {
        filter = f; border = withBorder ? f->borderSize : 0; sizeX = sx; sizeY = sy;
        data.assign((size_t) (sx + 2 * border) * (sy + 2 * border) * 5, 0.0f);
        int n = (int) std::ceil(2 * f->radius) + 1;
        weightsX.resize(n); weightsY.resize(n);
    }
    int bw() const { return sizeX + 2 * border; }
    int bh() const { return sizeY + 2 * border; }
    void clear() { std::fill(data.begin(), data.end(), 0.0f); }

    /* imageblock.h:124-204 */
    bool put(const Vec2 &_pos, const Float *value) {
        for (int i = 0; i < 5; ++i)
            if (!std::isfinite(value[i]) || value[i] < 0)
                return false;
        const Float filterRadius = filter->radius;
        const int sx = bw(), sy = bh();
        const Vec2 pos(_pos.x - 0.5f - (offX - border), _pos.y - 0.5f - (offY - border));
        const int minx = std::max((int) std::ceil(pos.x - filterRadius), 0),
                  miny = std::max((int) std::ceil(pos.y - filterRadius), 0),
                  maxx = std::min((int) std::floor(pos.x + filterRadius), sx - 1),
                  maxy = std::min((int) std::floor(pos.y + filterRadius), sy - 1);
        for (int x = minx, idx = 0; x <= maxx; ++x) weightsX[idx++] = filter->evalDiscretized(x - pos.x);
        for (int y = miny, idx = 0; y <= maxy; ++y) weightsY[idx++] = filter->evalDiscretized(y - pos.y);
        for (int y = miny, yr = 0; y <= maxy; ++y, ++yr) {
            const Float weightY = weightsY[yr];
            Float *dest = &data[((size_t) y * sx + minx) * 5];
            for (int x = minx, xr = 0; x <= maxx; ++x, ++xr) {
                const Float weight = weightsX[xr] * weightY;
                for (int k = 0; k < 5; ++k)
                    *dest++ += weight * value[k];
            }
        }
        return true;
    }

    /* imageblock.h:103-107 + bitmap.cpp:630-700 (accumulate with clipping) */
    void putBlock(const ImageBlock &b) {
        const int ox = b.offX - offX - (b.border - border), oy = b.offY - offY - (b.border - border);
        for (int y = 0; y < b.bh(); ++y) {
            int ty = y + oy; if (ty < 0 || ty >= bh()) continue;
            for (int x = 0; x < b.bw(); ++x) {
                int tx = x + ox; if (tx < 0 || tx >= bw()) continue;
                const Float *s = &b.data[((size_t) y * b.bw() + x) * 5];
                Float *d = &data[((size_t) ty * bw() + tx) * 5];
                for (int k = 0; k < 5; ++k) d[k] += s[k];
            }
        }
    }
};

/* imageproc.cpp:28-78 */
inline void spiralBlocks(int sizeX, int sizeY, int blockSize, std::vector<std::array<int, 4>> &out) {
    int nbx = (int) std::ceil((Float) sizeX / (Float) blockSize), nby = (int) std::ceil((Float) sizeY / (Float) blockSize);
    int total = nbx * nby, generated = 0;
    int cx = nbx / 2, cy = nby / 2, direction = 0 /* ERight */, stepsLeft = 1, numSteps = 1;
    out.clear();
    while (generated < total) {
        int px = cx * blockSize, py = cy * blockSize;
        out.push_back({ px, py, std::min(sizeX - px, blockSize), std::min(sizeY - py, blockSize) });
        if (++generated == total) break;
        do {
            switch (direction) { case 0: ++cx; break; case 1: ++cy; break; case 2: --cx; break; case 3: --cy; break; }
            if (--stepsLeft == 0) {
                direction = (direction + 1) % 4;
                if (direction == 2 || direction == 0) ++numSteps;
                stepsLeft = numSteps;
            }
        } while (cx < 0 || cy < 0 || cx >= nbx || cy >= nby);
    }
}

struct RenderParams {
    int spp = 4, blockSize = 32, threads = 1;
    IntegratorParams ip;
    bool direct = false;                 /* MIDirectIntegrator instead of MIPathTracer */
    bool volpath = false;                /* SimpleVolumetricPathTracer on a media-free scene (o_volpath.h) */
    DirectParams dp;
    bool ctr = true; uint32_t seed = 0;
    int shardIndex = 0, shardCount = 1;
    bool ld = false;                         /* PHIP_SAMPLER_LD on top of the counter stream */
    int qmc = 0; SobolTables sobol; uint32_t stRes = 1;      /* PHIP_SAMPLER_SOBOL (1) / PHIP_SAMPLER_STRATIFIED (2) / _HALTON, _HAMMERSLEY (3), `path` only */
    RinvTables rinv;
    int sampleOffset = 0, sampleTotal = 0;   /* phip_render_params::sample_offset / sample_total: this call renders samples [offset, offset + spp) of sampleTotal */
};

struct RenderResult {
    PathCounters counters;
    double seconds = 0;
};

/*
 * Renders the crop window into film (cropH*cropW*5 floats).  If sampleOut != nullptr it
 * receives cropW*cropH*spp (R,G,B,alpha) per-sample radiance values ordered [y][x][sample]; maskOut (optional, same order): per
 * sample, bit d-1 = the BSDF at path vertex d is smooth (what a parity-stream sampler for the reference needs to know).
 */
inline RenderResult render(const Scene &scene, const RenderParams &rp, float *filmOut, float *sampleOut, uint32_t *maskOut = nullptr) {
    const phip_film &f = scene.film;
    PerspectiveCamera cam; cam.configure(scene.camera, f);
    Filter filter; filter.configure(f);
    std::vector<std::array<int, 4>> blocks;
    spiralBlocks(f.crop_width, f.crop_height, rp.blockSize, blocks);

    /* renderproc.cpp:160-173: block offsets are relative to the crop window (offset 0,0, size = cropSize) */
    ImageBlock film; film.init(f.crop_width, f.crop_height, &filter, false);

    const int nThreads = std::max(1, rp.threads);
    std::vector<ImageBlock> results(blocks.size());
    std::vector<PathCounters> counters(nThreads);
    std::atomic<size_t> next(0);

    /* renderjob.cpp:58-69 + independent.cpp:71-80: one sampler clone per worker */
    SFMT parent(5489ULL);
    std::vector<SFMT> workerRng;
    if (!rp.ctr) { workerRng.resize(nThreads); for (int i = 0; i < nThreads; ++i) workerRng[i].seedFrom(parent); }

    const Float diffScaleFactor = 1.0f / std::sqrt((Float) (rp.sampleTotal > 0 ? rp.sampleTotal : rp.spp));      /* integrator.cpp:144-145 */
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        PathCounters &pc = counters[tid];
        for (;;) {
            size_t bi = next.fetch_add(1);
            if (bi >= blocks.size()) break;
            if (rp.shardCount > 1 && (int) (bi % (size_t) rp.shardCount) != rp.shardIndex) continue;
            const auto &b = blocks[bi];
            ImageBlock &blk = results[bi];
            blk.init(b[2], b[3], &filter, true);
            blk.offX = b[0]; blk.offY = b[1];
            SampleSource smp; smp.ctr = rp.ctr; smp.seed = rp.seed; smp.rng = rp.ctr ? nullptr : &workerRng[tid];
            smp.ld = rp.ld; smp.ldMask = (uint32_t) (rp.sampleTotal > 0 ? rp.sampleTotal : rp.spp) - 1u; smp.rrDepth = rp.ip.rrDepth;
            smp.qmc = rp.qmc; smp.sobol = &rp.sobol; smp.stRes = rp.stRes; smp.rinv = &rp.rinv;
            for (int y = 0; y < b[3]; ++y) for (int x = 0; x < b[2]; ++x) {
                const int px = blk.offX + x, py = blk.offY + y;     /* crop-window pixel coordinates */
                smp.pixel = (uint32_t) (py * f.crop_width + px);
                smp.px = (uint32_t) px; smp.py = (uint32_t) py;      /* sampler->generate(offset), integrator.cpp:164 (crop window at the film's origin) */
                if (rp.direct) smp.generateDirectArrays((size_t) rp.spp, rp.dp.emitterSamples, rp.dp.bsdfSamples);   /* sampler->generate(offset), integrator.cpp:164 */
                for (int j = 0; j < rp.spp; ++j) {
                    smp.sample = (uint32_t) (j + rp.sampleOffset);
                    Vec2 jit = smp.cameraSample();
                    Vec2 samplePos((Float) px + jit.x, (Float) py + jit.y);   /* integrator.cpp:171 */
                    Vec3 rx, ry;
                    Ray ray = cam.sampleRay(samplePos, &rx, &ry);
                    /* RayDifferential::scaleDifferential, ray.h:163-168, integrator.cpp:144-145,181 */
                    rx = ray.d + (rx - ray.d) * diffScaleFactor;
                    ry = ray.d + (ry - ray.d) * diffScaleFactor;
                    Float alpha;
                    pc.smoothMask = 0;
                    Spectrum spec = rp.direct ? directLi(scene, rp.dp, ray, smp, alpha, &pc, rx, ry)
                                  : rp.volpath ? volpathSimpleLi(scene, rp.ip, ray, smp, alpha, &pc, &rx, &ry)
                                               : pathLi(scene, rp.ip, ray, smp, alpha, &pc, &rx, &ry);
                    Float temp[5] = { spec[0], spec[1], spec[2], alpha, 1.0f };
                    if (!blk.put(samplePos, temp)) pc.invalidSamples++;
                    if (maskOut) maskOut[((size_t) py * f.crop_width + px) * rp.spp + j] = pc.smoothMask;
                    if (sampleOut) {
                        size_t si = (((size_t) py * f.crop_width + px) * rp.spp + j) * 4;
                        sampleOut[si] = spec[0]; sampleOut[si + 1] = spec[1]; sampleOut[si + 2] = spec[2]; sampleOut[si + 3] = alpha;
                    }
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nThreads; ++i) th.emplace_back(worker, i);
    worker(0);
    for (auto &t : th) t.join();
    auto t1 = std::chrono::steady_clock::now();

    /* film merge in block order (renderproc.cpp:142-149 does it under a mutex in completion order) */
    for (size_t bi = 0; bi < blocks.size(); ++bi)
        if (!results[bi].data.empty()) film.putBlock(results[bi]);
    memcpy(filmOut, film.data.data(), film.data.size() * sizeof(Float));

    RenderResult rr;
    for (auto &c : counters) rr.counters.add(c);
    rr.seconds = std::chrono::duration<double>(t1 - t0).count();
    return rr;
}

} // namespace orc
