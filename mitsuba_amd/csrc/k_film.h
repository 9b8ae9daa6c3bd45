/*
 * k_film.h -- statistics reduction, film gather (ImageBlock::put), per-sample export
 * Included by phip.hip; see its header for the kernel overview.
 */

/* sums the per-wave statistics: REDUCE_SPLIT blocks per counter row, rows [firstRow, firstRow + gridDim.x);
   the totals must have been zeroed (one atomicAdd per block: 32 per row) */
#define REDUCE_SPLIT 32
__global__ void k_reduce_stats(PathPool P, Counters *C, int firstRow) {
    __shared__ unsigned long long red[256];
    const int row = firstRow + (int) blockIdx.x;
    const unsigned long long *src = P.stat + (size_t) row * P.nWaves;
    unsigned long long v = 0;
    for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < P.nWaves; i += 256 * REDUCE_SPLIT) v += src[i];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int) threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0 && red[0]) atomicAdd(&C->total[row], red[0]);
}

/* Film: one lane per crop pixel gathers every sample whose filter footprint covers it.  Restates
 * ImageBlock::put (imageblock.h:124-204) incl. the block-local coordinate arithmetic: a sample
 * taken in pixel (sx,sy) belongs to the render block whose origin is (sx,sy) rounded down to the
 * block size, and its weights are computed in that block's coordinate system. */
template <bool QMC>        /* QMC: the sample positions of PHIP_SAMPLER_SOBOL / _STRATIFIED (the tiled kernels serve the counter and LD streams) */
__global__ __launch_bounds__(BLOCK) void k_film(DevScene S, RenderConst rc, const float4 *L, const int32_t *tileSlot,
                                               int tilesX, float *out, int accumulate, unsigned long long *invalidCount) {
    const DevFilm &F = S.film;
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= F.width || y >= F.height) return;
    /* a sample of source pixel s lands at s + jitter - 0.5 in [s-0.5, s+0.5): it can reach x iff
       s > x - radius - 0.5 and s <= x + radius + 0.5 */
    const int sx0 = max((int) floorf((float) x - F.radius - 0.5f) + 1, 0), sx1 = min((int) floorf((float) x + F.radius + 0.5f), F.width - 1);
    const int sy0 = max((int) floorf((float) y - F.radius - 0.5f) + 1, 0), sy1 = min((int) floorf((float) y + F.radius + 0.5f), F.height - 1);
    float acc[5] = { 0, 0, 0, 0, 0 };
    unsigned long long invalid = 0;
    for (int sy = sy0; sy <= sy1; ++sy) {
        for (int sx = sx0; sx <= sx1; ++sx) {
            const int tx = sx >> rc.tileShift, ty = sy >> rc.tileShift;
            const int32_t ts = tileSlot[ty * tilesX + tx];
            if (ts < 0) continue;           /* that block belongs to another shard */
            const int offX = tx << rc.tileShift, offY = ty << rc.tileShift;
            const int bw = min(F.blockSize, F.width - offX) + 2 * F.border, bh = min(F.blockSize, F.height - offY) + 2 * F.border;
            /* destination pixel in the source block's bitmap coordinates */
            const int dx = x - (offX - F.border), dy = y - (offY - F.border);
            if (dx < 0 || dy < 0 || dx >= bw || dy >= bh) continue;
            const uint32_t m = spreadBits((uint32_t) (sx - offX)) | (spreadBits((uint32_t) (sy - offY)) << 1);
            const uint32_t pixel = (uint32_t) sy * (uint32_t) F.width + (uint32_t) sx;
            for (uint32_t k = 0; k < rc.sppPass; ++k) {
                const unsigned long long id = (((unsigned long long) ts * rc.sppPass + k) << (2 * rc.tileShift)) | m;
                const V2 jit = filmJitter<QMC>(rc, id, pixel, k + rc.sppFirst, (uint32_t) F.width);
                const float px = (float) sx + jit.x, py = (float) sy + jit.y;
                const float posx = px - 0.5f - (float) (offX - F.border), posy = py - 0.5f - (float) (offY - F.border);
                const int minx = max((int) ceilf(posx - F.radius), 0), maxx = min((int) floorf(posx + F.radius), bw - 1);
                const int miny = max((int) ceilf(posy - F.radius), 0), maxy = min((int) floorf(posy + F.radius), bh - 1);
                if (dx < minx || dx > maxx || dy < miny || dy > maxy) continue;
                const float4 v = L[id];
                /* validity check of ImageBlock::put: reject non-finite / negative samples (imageblock.h:148-151) */
                if (!(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w)) || v.x < 0 || v.y < 0 || v.z < 0 || v.w < 0) {
                    if (sx == x && sy == y) ++invalid;
                    continue;
                }
                const float wx = F.table[min((int) fabsf(((float) dx - posx) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                const float wy = F.table[min((int) fabsf(((float) dy - posy) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                const float w = wx * wy;
                acc[0] += w * v.x; acc[1] += w * v.y; acc[2] += w * v.z; acc[3] += w * v.w; acc[4] += w * 1.0f;
            }
        }
    }
    float *o = out + ((size_t) y * F.width + x) * 5;
    if (accumulate) { for (int i = 0; i < 5; ++i) o[i] += acc[i]; }
    else { for (int i = 0; i < 5; ++i) o[i] = acc[i]; }
    if (invalid) atomicAdd(invalidCount, invalid);
}

/* LDS-tiled film gather (filters with reach <= FILM_MAX_REACH pixels, i.e. every reference default): a block owns
 * 16x16 destination pixels.  Per sample index k the block first STAGES each of the (16+2R)^2 source pixels' sample
 * ONCE in LDS: radiance, the frame coordinates of the first pixel of its filter footprint and the separable filter
 * weights of ImageBlock::put (imageblock.h:124-204: footprint clipped to the bitmap of the render block the sample
 * belongs to, weights from the discretised table) -- weights outside the footprint are stored as 0, which adds
 * nothing.  Then every destination lane accumulates its (2R+1)^2 neighbours: two integer subtractions, two weight
 * reads, one product and five multiply-adds per neighbour.  Same arithmetic per (sample, pixel) pair as k_film;
 * only the order of the float additions differs. */
#define FILM_MAX_REACH 4
#define FILM_TILE 16
template <int RMAX>
__global__ __launch_bounds__(BLOCK) void k_film_tiled(DevScene S, RenderConst rc, const float4 *L, const int32_t *tileSlot,
                                                     int tilesX, float *out, int accumulate, unsigned long long *invalidCount, int R) {
    constexpr int TMAX = FILM_TILE + 2 * RMAX;
    constexpr int NW = 2 * RMAX + 2;           /* weights per axis: floor(p + r) - ceil(p - r) + 1 <= 2r + 1 with r < RMAX + 0.5 */
    __shared__ float4 sVal[TMAX * TMAX];       /* radiance rgb + alpha of the source pixel's k-th sample */
    __shared__ int2 sOrg[TMAX * TMAX];         /* frame coordinates of weight [0] of the sample's footprint */
    __shared__ float sWx[TMAX * TMAX * NW], sWy[TMAX * TMAX * NW];
    __shared__ int4 sGeo[TMAX * TMAX];         /* (offX - border, offY - border, bw, bh) of the source pixel's render block */
    __shared__ uint32_t sBase[TMAX * TMAX];    /* low word of the sample id of k = 0 (0xFFFFFFFF: pixel not rendered here) */
    __shared__ float sTable[PHIP_FILTER_RESOLUTION + 1];
    const DevFilm &F = S.film;
    const int T = FILM_TILE + 2 * R;
    const int x0 = blockIdx.x * FILM_TILE, y0 = blockIdx.y * FILM_TILE;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x = x0 + lx, y = y0 + ly;
    if (threadIdx.x <= PHIP_FILTER_RESOLUTION) sTable[threadIdx.x] = F.table[threadIdx.x];
    /* per source pixel constants */
    for (int i = threadIdx.x; i < T * T; i += BLOCK) {
        const int sx = x0 - R + (i % T), sy = y0 - R + (i / T);
        uint32_t base = 0xFFFFFFFFu; int4 g = make_int4(0, 0, 0, 0);
        if (sx >= 0 && sy >= 0 && sx < F.width && sy < F.height) {
            const int tx = sx >> rc.tileShift, ty = sy >> rc.tileShift;
            const int32_t ts = tileSlot[ty * tilesX + tx];
            if (ts >= 0) {
                const int offX = tx << rc.tileShift, offY = ty << rc.tileShift;
                g = make_int4(offX - F.border, offY - F.border, min(F.blockSize, F.width - offX) + 2 * F.border, min(F.blockSize, F.height - offY) + 2 * F.border);
                base = (uint32_t) ts;                       /* id(k) = ((ts * sppPass + k) << 2*tileShift) | morton(pixel in block) */
            }
        }
        sBase[i] = base; sGeo[i] = g;
    }
    __syncthreads();

    float acc[5] = { 0, 0, 0, 0, 0 };
    unsigned long long invalid = 0;
    const bool inside = x < F.width && y < F.height;
    /* the radiance of sample k + 1 is fetched while sample k is being gathered (the k loop is a chain of
       barriers otherwise: global-load latency would be paid sppPass times in a row) */
    constexpr int NSTAGE = (TMAX * TMAX + BLOCK - 1) / BLOCK;
    float4 pre[NSTAGE];
    auto fetch = [&](uint32_t k) {
#pragma unroll
        for (int n = 0; n < NSTAGE; ++n) {
            const int i = (int) threadIdx.x + n * BLOCK;
            pre[n] = make_float4(0, 0, 0, 0);
            if (i < T * T && k < rc.sppPass) {
                const uint32_t base = sBase[i];
                if (base != 0xFFFFFFFFu) {
                    const int sx = x0 - R + (i % T), sy = y0 - R + (i / T);
                    const int4 g = sGeo[i];
                    const uint32_t m = spreadBits((uint32_t) (sx - (g.x + F.border))) | (spreadBits((uint32_t) (sy - (g.y + F.border))) << 1);
                    pre[n] = L[(((unsigned long long) base * rc.sppPass + k) << (2 * rc.tileShift)) | m];
                }
            }
        }
    };
    fetch(0);
    for (uint32_t k = 0; k < rc.sppPass; ++k) {
#pragma unroll
        for (int n = 0; n < NSTAGE; ++n) {
            const int i = (int) threadIdx.x + n * BLOCK;
            if (i >= T * T) break;
            const uint32_t base = sBase[i];
            float4 v = make_float4(0, 0, 0, 0);
            int2 org = make_int2(0, 0);
            float wx[NW], wy[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) { wx[j] = 0.0f; wy[j] = 0.0f; }
            if (base != 0xFFFFFFFFu) {
                const int sx = x0 - R + (i % T), sy = y0 - R + (i / T);
                const int4 g = sGeo[i];
                const uint32_t pixel = (uint32_t) sy * (uint32_t) F.width + (uint32_t) sx;
                const V2 jit = streamJitter(rc, pixel, k + rc.sppFirst);
                const float px = (float) sx + jit.x, py = (float) sy + jit.y;
                const float posx = px - 0.5f - (float) g.x, posy = py - 0.5f - (float) g.y;   /* block-bitmap coordinates */
                v = pre[n];
                /* validity check of ImageBlock::put (imageblock.h:148-151) */
                if (!(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w)) || v.x < 0 || v.y < 0 || v.z < 0 || v.w < 0) {
                    /* count each rejected sample once: by the block that owns its pixel */
                    if (sx >= x0 && sx < x0 + FILM_TILE && sy >= y0 && sy < y0 + FILM_TILE) ++invalid;
                    v = make_float4(0, 0, 0, 0);
                } else {
                    /* footprint and weights, imageblock.h:159-180 */
                    const int uminx = (int) ceilf(posx - F.radius), uminy = (int) ceilf(posy - F.radius);
                    const int minx = max(uminx, 0), maxx = min((int) floorf(posx + F.radius), g.z - 1);
                    const int miny = max(uminy, 0), maxy = min((int) floorf(posy + F.radius), g.w - 1);
                    org = make_int2(g.x + uminx, g.y + uminy);
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const int bx = uminx + j, by = uminy + j;
                        if (bx >= minx && bx <= maxx) wx[j] = sTable[min((int) fabsf(((float) bx - posx) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                        if (by >= miny && by <= maxy) wy[j] = sTable[min((int) fabsf(((float) by - posy) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                    }
                }
            }
            sVal[i] = v; sOrg[i] = org;
#pragma unroll
            for (int j = 0; j < NW; ++j) { sWx[i * NW + j] = wx[j]; sWy[i * NW + j] = wy[j]; }
        }
        __syncthreads();
        fetch(k + 1);
        if (inside) {
            for (int dyy = -R; dyy <= R; ++dyy) {
                for (int dxx = -R; dxx <= R; ++dxx) {
                    const int i = (ly + R + dyy) * T + (lx + R + dxx);
                    const int2 org = sOrg[i];
                    const int jx = x - org.x, jy = y - org.y;
                    if ((unsigned) jx >= (unsigned) NW || (unsigned) jy >= (unsigned) NW) continue;
                    const float w = sWx[i * NW + jx] * sWy[i * NW + jy];
                    if (w == 0.0f) continue;                   /* outside the footprint (or a zero of the filter): adds nothing */
                    const float4 v = sVal[i];
                    acc[0] += w * v.x; acc[1] += w * v.y; acc[2] += w * v.z; acc[3] += w * v.w; acc[4] += w * 1.0f;
                }
            }
        }
        __syncthreads();
    }
    if (inside) {
        float *o = out + ((size_t) y * F.width + x) * 5;
        if (accumulate) { for (int i = 0; i < 5; ++i) o[i] += acc[i]; }
        else { for (int i = 0; i < 5; ++i) o[i] = acc[i]; }
    }
    if (invalid) atomicAdd(invalidCount, invalid);
}

#if PHIP_EXPERIMENTS      /* (superseded by the splat below for every case it served; kept for A/B builds: PHIP_FILM_GATHER=1) */
/* ---- round 3: the tiled gather for the reference's default filters (reach R = 1 or 2 pixels: box, tent, gaussian stddev 0.5) ----
 * Same arithmetic per (sample, pixel) pair and the same order of additions as k_film_tiled, restructured around what round 2's
 * counters showed (53 % of the wave cycles waiting at 3.7 waves per SIMD, 37 KB of LDS per block, two barriers per sample index):
 *   - a sample's footprint starts one or two pixels left of its own pixel (ceil(jitter - 0.5 - radius)), so its separable weights are
 *     staged at ABSOLUTE offsets, wx[j] = weight of pixel sx - R + j (0 outside the footprint or the render block's bitmap): the
 *     gather needs no footprint origin, no bounds test and no branch -- w = wx[R - dx] * wy[R - dy] for all (2R+1)^2 neighbours,
 *     fully unrolled (a zero weight adds +0: same bits as skipping it, invalid samples are staged as zero radiance);
 *   - staging is double-buffered: sample k + 1 is staged while sample k is gathered, ONE barrier per sample index;
 *   - per source pixel constants shrink to the sample-id base and an index into the (at most four) render blocks the tile touches.
 * LDS: 2 x (20x20) x (16 + 44) B + 3.3 KB = 51 KB per block for R = 2. */
#ifndef FILM_NBUF
#define FILM_NBUF 2                  /* staging buffers of k_film_tiled2: 2 = sample k + 1 is staged while k is gathered (one barrier per sample index, 51 KB: 3 blocks per CU);
                                        1 = stage, barrier, gather, barrier (26 KB: 6 blocks per CU) */
#endif
template <int R>
__global__ __launch_bounds__(BLOCK) void k_film_tiled2(DevScene S, RenderConst rc, const float4 *L, const int32_t *tileSlot,
                                                      int tilesX, float *out, int accumulate, unsigned long long *invalidCount) {
    constexpr int T = FILM_TILE + 2 * R, NW = 2 * R + 1, NP = T * T;
    constexpr int NSTAGE = (NP + BLOCK - 1) / BLOCK;
    __shared__ float4 sVal[FILM_NBUF][NP];
    __shared__ float sW[FILM_NBUF][NP][2 * NW + 1];        /* wx[0..NW), wy[0..NW) at absolute offsets (+1: an odd stride keeps consecutive source pixels on different banks) */
    __shared__ uint32_t sBase[NP];                 /* local tile slot of the source pixel's render block (0xFFFFFFFF: not rendered here) */
    __shared__ uint32_t sMorton[NP];               /* Morton index of the source pixel in its block | geometry index << 30 */
    __shared__ int4 sGeo[4];                       /* (offX - border, offY - border, bw, bh) of the up to 2 x 2 render blocks under the tile */
    __shared__ float sTable[PHIP_FILTER_RESOLUTION + 1];
    const DevFilm &F = S.film;
    const int x0 = blockIdx.x * FILM_TILE, y0 = blockIdx.y * FILM_TILE;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x = x0 + lx, y = y0 + ly;
    if (threadIdx.x <= PHIP_FILTER_RESOLUTION) sTable[threadIdx.x] = F.table[threadIdx.x];
    /* the render blocks under the tile + halo: block size >= 2 R + ... >= FILM_TILE is NOT assumed -- at most 2 x 2 blocks when
       blockSize >= T, which the host guarantees before choosing this kernel */
    const int tx0 = max(x0 - R, 0) >> rc.tileShift, ty0 = max(y0 - R, 0) >> rc.tileShift;
    if (threadIdx.x < 4) {
        const int tx = tx0 + (threadIdx.x & 1), ty = ty0 + (threadIdx.x >> 1);
        const int offX = tx << rc.tileShift, offY = ty << rc.tileShift;
        sGeo[threadIdx.x] = make_int4(offX - F.border, offY - F.border, min(F.blockSize, F.width - offX) + 2 * F.border, min(F.blockSize, F.height - offY) + 2 * F.border);
    }
    for (int i = threadIdx.x; i < NP; i += BLOCK) {
        const int sx = x0 - R + (i % T), sy = y0 - R + (i / T);
        uint32_t base = 0xFFFFFFFFu, mm = 0;
        if (sx >= 0 && sy >= 0 && sx < F.width && sy < F.height) {
            const int tx = sx >> rc.tileShift, ty = sy >> rc.tileShift;
            const int32_t ts = tileSlot[ty * tilesX + tx];
            if (ts >= 0) {
                base = (uint32_t) ts;
                mm = (spreadBits((uint32_t) (sx - (tx << rc.tileShift))) | (spreadBits((uint32_t) (sy - (ty << rc.tileShift))) << 1))
                   | ((uint32_t) ((tx - tx0) | ((ty - ty0) << 1)) << 30);
            }
        }
        sBase[i] = base; sMorton[i] = mm;
    }
    __syncthreads();

    float acc[5] = { 0, 0, 0, 0, 0 };
    unsigned long long invalid = 0;
    const bool inside = x < F.width && y < F.height;
    float4 pre[NSTAGE];
    auto fetch = [&](uint32_t k) {
#pragma unroll
        for (int n = 0; n < NSTAGE; ++n) {
            const int i = (int) threadIdx.x + n * BLOCK;
            pre[n] = make_float4(0, 0, 0, 0);
            if (i < NP && k < rc.sppPass) {
                const uint32_t base = sBase[i];
                if (base != 0xFFFFFFFFu)
                    pre[n] = L[(((unsigned long long) base * rc.sppPass + k) << (2 * rc.tileShift)) | (sMorton[i] & 0x3FFFFFFFu)];
            }
        }
    };
    auto stage = [&](uint32_t k, int buf) {
#pragma unroll
        for (int n = 0; n < NSTAGE; ++n) {
            const int i = (int) threadIdx.x + n * BLOCK;
            if (i >= NP) break;
            float4 v = make_float4(0, 0, 0, 0);
            float w[2 * NW];
#pragma unroll
            for (int j = 0; j < 2 * NW; ++j) w[j] = 0.0f;
            if (sBase[i] != 0xFFFFFFFFu) {
                const int sx = x0 - R + (i % T), sy = y0 - R + (i / T);
                const int4 g = sGeo[sMorton[i] >> 30];
                const uint32_t pixel = (uint32_t) sy * (uint32_t) F.width + (uint32_t) sx;
                const V2 jit = streamJitter(rc, pixel, k + rc.sppFirst);
                const float px = (float) sx + jit.x, py = (float) sy + jit.y;
                const float posx = px - 0.5f - (float) g.x, posy = py - 0.5f - (float) g.y;   /* block-bitmap coordinates */
                v = pre[n];
                /* validity check of ImageBlock::put (imageblock.h:148-151) */
                if (!(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w)) || v.x < 0 || v.y < 0 || v.z < 0 || v.w < 0) {
                    if (sx >= x0 && sx < x0 + FILM_TILE && sy >= y0 && sy < y0 + FILM_TILE) ++invalid;      /* counted once: by the tile that owns the pixel */
                    v = make_float4(0, 0, 0, 0);
                } else {
                    /* footprint and weights, imageblock.h:159-180 */
                    const int minx = max((int) ceilf(posx - F.radius), 0), maxx = min((int) floorf(posx + F.radius), g.z - 1);
                    const int miny = max((int) ceilf(posy - F.radius), 0), maxy = min((int) floorf(posy + F.radius), g.w - 1);
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const int bx = sx - R + j - g.x, by = sy - R + j - g.y;             /* bitmap coordinates of the pixels sx - R + j, sy - R + j */
                        if (bx >= minx && bx <= maxx) w[j] = sTable[min((int) fabsf(((float) bx - posx) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                        if (by >= miny && by <= maxy) w[NW + j] = sTable[min((int) fabsf(((float) by - posy) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                    }
                }
            }
            sVal[buf][i] = v;
#pragma unroll
            for (int j = 0; j < 2 * NW; ++j) sW[buf][i][j] = w[j];
        }
    };
    fetch(0);
    if (FILM_NBUF == 2) {
        if (rc.sppPass) stage(0, 0);
        fetch(1);
        __syncthreads();
    }
    for (uint32_t k = 0; k < rc.sppPass; ++k) {
        const int buf = FILM_NBUF == 2 ? (int) (k & 1u) : 0;
        if (FILM_NBUF == 2) {
            if (k + 1 < rc.sppPass) stage(k + 1, buf ^ 1);  /* (uses pre[] = sample k + 1; the other buffer) */
            fetch(k + 2);
        } else {
            stage(k, 0);
            __syncthreads();
            fetch(k + 1);
        }
        if (inside) {
#pragma unroll
            for (int dyy = -R; dyy <= R; ++dyy) {
#pragma unroll
                for (int dxx = -R; dxx <= R; ++dxx) {
                    const int i = (ly + R + dyy) * T + (lx + R + dxx);
                    const float w = sW[buf][i][R - dxx] * sW[buf][i][NW + R - dyy];
                    const float4 v = sVal[buf][i];
                    acc[0] += w * v.x; acc[1] += w * v.y; acc[2] += w * v.z; acc[3] += w * v.w; acc[4] += w * 1.0f;
                }
            }
        }
        __syncthreads();
    }
    if (inside) {
        float *o = out + ((size_t) y * F.width + x) * 5;
        if (accumulate) { for (int i = 0; i < 5; ++i) o[i] += acc[i]; }
        else { for (int i = 0; i < 5; ++i) o[i] = acc[i]; }
    }
    if (invalid) atomicAdd(invalidCount, invalid);
}

#endif  /* PHIP_EXPERIMENTS */

/* ---- round 4: the film as a SPLAT in registers + an ordered merge (filters of reach R <= 2: every default of the reference) ----
 * The gathers above read each staged sample 25 times out of LDS (one per destination pixel under its footprint), stage 1.56 source
 * pixels per destination pixel (the halo), and synchronise the block once per sample index: 5.0 ms per C2 frame at 0.17 of the HBM
 * rate for what is one pass over 4.3 GB.  Here a thread OWNS a source pixel for the whole pass:
 *   k_film_splat   block = one 16 x 16 patch of a render block (256 consecutive Morton indices: L is read in 1 KB runs, every sample once),
 *                  thread = one source pixel; it walks the pixel's samples and keeps the (2R+1)^2 x 5 partial sums of its footprint in
 *                  REGISTERS (125 accumulators for R = 2: jitter and ten table weights once per sample, then 25 products and 125 fused
 *                  multiply-adds -- no LDS, no barrier inside the pass).  At the end the block folds its threads' partial sums into a
 *                  (16+2R)^2 x 5 patch image in LDS -- (2R+1)^2 rounds, in round j every thread adds its j-th partial to the cell at
 *                  offset j of its own pixel: distinct cells within a round, a fixed order over the rounds, so the sums are deterministic --
 *                  and stores the patch image (8 KB);
 *   k_film_merge   thread = one film pixel: adds the cells the (at most four of nine neighbouring) patch images hold for it, in a fixed order.
 * ImageBlock::put's arithmetic per (sample, pixel) pair is unchanged (imageblock.h:148-186: validity test, block-local position, footprint,
 * table weights, weight = wx * wy); the ORDER of the additions per film pixel is this kernel's own (per source pixel over its samples, then
 * over the source pixels) -- the reference's depends on its thread schedule.  Both device paths (k_mega, wavefront) share it. */
#ifndef FILM_SPLAT_PF
#define FILM_SPLAT_PF 4
#endif
#ifndef FILM_SPLAT_WAVES
#define FILM_SPLAT_WAVES 2                /* 125 accumulators + the working set of one sample: 256 VGPRs, two waves per SIMD (three: 168 VGPRs and 640 B of scratch) */
#endif
template <int R, bool QMC>
__global__ __launch_bounds__(256, FILM_SPLAT_WAVES) void k_film_splat(DevScene S, RenderConst rc, const float4 *L, float *patchImages, unsigned long long *invalidCount) {
    constexpr int NW = 2 * R + 1, T = 16 + 2 * R, NP = T * T;
    __shared__ float sImg[NP * 5];
    __shared__ float sTable[PHIP_FILTER_RESOLUTION + 1];
    const DevFilm &F = S.film;
    const uint32_t patchesPerTile = rc.tilePixels >> 8;
    const uint32_t ts = blockIdx.x / patchesPerTile, q = blockIdx.x - ts * patchesPerTile;          /* local render block, patch inside it */
    const uint32_t m = (q << 8) | threadIdx.x;                                                        /* Morton index of this thread's pixel in the render block */
    const uint32_t org = rc.tileOrigin[ts];
    const int offX = (int) (org & 0xFFFFu), offY = (int) (org >> 16);
    const int sxP = offX + (int) compactBits(m), syP = offY + (int) compactBits(m >> 1);
    const int lx = (int) compactBits(threadIdx.x), ly = (int) compactBits(threadIdx.x >> 1);        /* position inside the patch */
    if (threadIdx.x <= PHIP_FILTER_RESOLUTION) sTable[threadIdx.x] = F.table[threadIdx.x];
    for (int i = threadIdx.x; i < NP * 5; i += 256) sImg[i] = 0.0f;
    __syncthreads();
    const bool inside = sxP < F.width && syP < F.height;
    /* the render block's bitmap: (offX - border, offY - border), bw x bh (imageblock.cpp:26-30, renderproc.cpp:160-173) */
    const int gx = offX - F.border, gy = offY - F.border;
    const int bw = min(F.blockSize, F.width - offX) + 2 * F.border, bh = min(F.blockSize, F.height - offY) + 2 * F.border;
    f4v part[NW * NW]; float partW[NW * NW];                  /* (R, G, B, alpha) as register quads -- pairs for v_pk_fma_f32 without padding -- and the weight */
#pragma unroll
    for (int j = 0; j < NW * NW; ++j) { part[j] = f4v{ 0.0f, 0.0f, 0.0f, 0.0f }; partW[j] = 0.0f; }
    unsigned long long invalid = 0;
    if (inside) {
        const uint32_t pixelP = (uint32_t) syP * (uint32_t) F.width + (uint32_t) sxP;
        const float4 *src = L + ((((unsigned long long) ts * rc.sppPass) << (2 * rc.tileShift)) | m);
        const size_t stride = (size_t) 1 << (2 * rc.tileShift);
        constexpr uint32_t PF = FILM_SPLAT_PF;                                                                 /* samples in flight */
        float4 pre[PF];
#pragma unroll
        for (uint32_t i = 0; i < PF; ++i) pre[i] = i < rc.sppPass ? src[i * stride] : make_float4(0, 0, 0, 0);
        for (uint32_t k0 = 0; k0 < rc.sppPass; k0 += PF) {
#pragma unroll
            for (uint32_t i = 0; i < PF; ++i) {
                const uint32_t k = k0 + i;
                const float4 v = pre[i];
                if (k + PF < rc.sppPass) pre[i] = src[(size_t) (k + PF) * stride];
                const bool live = k < rc.sppPass;
                /* (the pixel's coordinates are made opaque per sample: everything derived from them -- ten bitmap columns / rows, their float forms, the
                   first rounds of the jitter hash -- is loop-invariant, and hoisted out of the loop it cost 35 registers the 125 accumulators need) */
                int sx = sxP, sy = syP; uint32_t pixel = pixelP;
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(sx), "+v"(sy), "+v"(pixel));
#endif
                /* (sequence samplers: the jitter the path kernel stored for this sample id, behind the sample itself; a padding sample reads the pass's first) */
                const V2 jit = filmJitter<QMC>(rc, (unsigned long long) (src - L) + (size_t) (live ? k : 0u) * stride, pixel, k + rc.sppFirst, (uint32_t) F.width);
                const float px = (float) sx + jit.x, py = (float) sy + jit.y;
                const float posx = px - 0.5f - (float) gx, posy = py - 0.5f - (float) gy;             /* block-bitmap coordinates */
                /* validity check of ImageBlock::put (imageblock.h:148-151).  No branch around the 125 accumulators (the two paths' copies of them do not fit
                   the register file): a rejected sample -- and the padding behind the pass's last sample -- is zero radiance under zero weights */
                const bool bad = !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w)) || v.x < 0 || v.y < 0 || v.z < 0 || v.w < 0;
                invalid += (live && bad) ? 1ull : 0ull;
                const bool ok = live && !bad;
                /* footprint and weights, imageblock.h:159-180; wx[j] / wy[j] = weight of pixel sx - R + j / sy - R + j (0 outside the footprint or the bitmap) */
                const int minx = max((int) ceilf(posx - F.radius), 0), maxx = min((int) floorf(posx + F.radius), bw - 1);
                const int miny = max((int) ceilf(posy - F.radius), 0), maxy = min((int) floorf(posy + F.radius), bh - 1);
                const f4v vv = { ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f };
                float wx[NW], wy[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    const int bx = sx - R + j - gx, by = sy - R + j - gy;
                    const float tx_ = sTable[min((int) fabsf(((float) bx - posx) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                    const float ty_ = sTable[min((int) fabsf(((float) by - posy) * F.scaleFactor), PHIP_FILTER_RESOLUTION)];
                    wx[j] = (ok && bx >= minx && bx <= maxx) ? tx_ : 0.0f;
                    wy[j] = (by >= miny && by <= maxy) ? ty_ : 0.0f;
                }
#pragma unroll
                for (int jy = 0; jy < NW; ++jy) {
#pragma unroll
                    for (int jx = 0; jx < NW; ++jx) {
                        const float w = wx[jx] * wy[jy];
                        part[jy * NW + jx] = __builtin_elementwise_fma(f4v{ w, w, w, w }, vv, part[jy * NW + jx]);
                        partW[jy * NW + jx] += w;
                    }
                }
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);               /* one sample after the other: interleaved, the unrolled copies' temporaries pushed the accumulators out of the registers */
#endif
            }
        }
    }
    /* fold the threads' partial sums into the patch image: round j = offset (jx, jy) of every thread's own pixel */
#pragma unroll
    for (int j = 0; j < NW * NW; ++j) {
        float *c = &sImg[((ly + j / NW) * T + (lx + j % NW)) * 5];
        c[0] += part[j].x; c[1] += part[j].y; c[2] += part[j].z; c[3] += part[j].w; c[4] += partW[j];
        __syncthreads();
    }
    float *dst = patchImages + (size_t) blockIdx.x * (NP * 5);
    for (int i = threadIdx.x; i < NP * 5; i += 256) dst[i] = sImg[i];
    if (invalid) atomicAdd(invalidCount, invalid);
}

/* thread = one film pixel: the sum of what the patch images around it hold for it (its own patch and those of the eight neighbouring patches
   whose halo reaches it), own patch first, then row by row */
template <int R>
__global__ __launch_bounds__(256) void k_film_merge(DevScene S, RenderConst rc, const float *patchImages, const int32_t *tileSlot, int tilesX, float *out, int accumulate) {
    constexpr int T = 16 + 2 * R, NP = T * T;
    const DevFilm &F = S.film;
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= F.width || y >= F.height) return;
    const int PX = x >> 4, PY = y >> 4, patchShift = (int) rc.tileShift - 4;                          /* patches per render-block side = 1 << patchShift */
    float acc[5] = { 0, 0, 0, 0, 0 };
#pragma unroll
    for (int o = 0; o < 9; ++o) {
        const int dx = o == 0 ? 0 : ((o - (o <= 4 ? 1 : 0)) % 3) - 1, dy = o == 0 ? 0 : ((o - (o <= 4 ? 1 : 0)) / 3) - 1;      /* (0,0) first, then the other eight row by row */
        const int qx = PX + dx, qy = PY + dy;
        if (qx < 0 || qy < 0 || (qx << 4) >= F.width || (qy << 4) >= F.height) continue;
        const int cx = x - (qx << 4) + R, cy = y - (qy << 4) + R;                                    /* cell of (x, y) in that patch's image */
        if (cx < 0 || cy < 0 || cx >= T || cy >= T) continue;
        const int tx = qx >> patchShift, ty = qy >> patchShift;
        const int32_t ts = tileSlot[ty * tilesX + tx];
        if (ts < 0) continue;                                                                         /* that render block belongs to another shard */
        const uint32_t pq = (spreadBits((uint32_t) (qx - (tx << patchShift))) | (spreadBits((uint32_t) (qy - (ty << patchShift))) << 1));
        const float *c = patchImages + ((size_t) ((uint32_t) ts << (2 * patchShift)) + pq) * (NP * 5) + (size_t) (cy * T + cx) * 5;
        acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2]; acc[3] += c[3]; acc[4] += c[4];
    }
    float *o = out + ((size_t) y * F.width + x) * 5;
    if (accumulate) { for (int i = 0; i < 5; ++i) o[i] += acc[i]; }
    else { for (int i = 0; i < 5; ++i) o[i] = acc[i]; }
}

/* copy per-sample radiance out in [y][x][sample] order (tests) */
__global__ void k_export_samples(DevScene S, RenderConst rc, const float4 *L, const int32_t *tileSlot, int tilesX,
                                 float4 *out, uint32_t sppTotal, uint32_t sampleOffset /* phip_render_params::sample_offset: index of the call's first sample */) {
    const DevFilm &F = S.film;
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t) F.width * F.height * rc.sppPass;
    if (i >= n) return;
    const uint32_t k = (uint32_t) (i % rc.sppPass);
    const size_t p = i / rc.sppPass;
    const int x = (int) (p % F.width), y = (int) (p / F.width);
    const int tx = x >> rc.tileShift, ty = y >> rc.tileShift;
    const int32_t ts = tileSlot[ty * tilesX + tx];
    float4 v = make_float4(0, 0, 0, 0);
    if (ts >= 0) {
        const uint32_t m = spreadBits((uint32_t) (x - (tx << rc.tileShift))) | (spreadBits((uint32_t) (y - (ty << rc.tileShift))) << 1);
        v = L[(((unsigned long long) ts * rc.sppPass + k) << (2 * rc.tileShift)) | m];
    }
    out[p * sppTotal + (rc.sppFirst - sampleOffset) + k] = v;
}

