/*
 * phip_debug.inl -- unit-test entry points, compiled into libphip_debug.so only (-DPHIP_DEBUG_HOOKS=1; the product library does not carry them).
 *
 * The phip_debug_host_* functions execute the
 * __host__ __device__ shading functions of dv_math.h / dv_scene.h ON THE HOST so that
 * `-m "not gpu"` tests can compare the product's arithmetic with the oracle bit for bit
 * without a GPU.  They are never reachable from phip_render*: rendering has no CPU fallback.
 */
extern "C" {

int phip_debug_host_bsdf_sample(const phip_material *materials, uint32_t n_materials, uint32_t material, size_t n,
                                const float *wi3, const float *sample2, float *wo3, float *weight3, float *pdf, uint8_t *delta) {
    try {
        std::vector<DevMaterial> mats = convertMaterials(materials, n_materials);
        if (material >= n_materials) return setErr(PHIP_ERR_INVALID, "material id out of range");
        DevScene S; memset(&S, 0, sizeof(S)); S.materials = mats.data();
        for (size_t i = 0; i < n; ++i) {
            BSDFSample bs;
            V3 w = bsdfSample(S, mats[material], V3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), V2(sample2[2 * i], sample2[2 * i + 1]), bs);
            if (w.isZero()) { bs.pdf = 0; bs.wo = V3(0.0f); }
            wo3[3 * i] = bs.wo.x; wo3[3 * i + 1] = bs.wo.y; wo3[3 * i + 2] = bs.wo.z;
            weight3[3 * i] = w.x; weight3[3 * i + 1] = w.y; weight3[3 * i + 2] = w.z;
            pdf[i] = bs.pdf; if (delta) delta[i] = bs.delta ? 1 : 0;
        }
        return PHIP_OK;
    } catch (const std::exception &e) { return setErr(PHIP_ERR_INVALID, e.what()); }
}

int phip_debug_host_bsdf_eval_pdf(const phip_material *materials, uint32_t n_materials, uint32_t material, size_t n,
                                  const float *wi3, const float *wo3, float *value3, float *pdf) {
    try {
        std::vector<DevMaterial> mats = convertMaterials(materials, n_materials);
        if (material >= n_materials) return setErr(PHIP_ERR_INVALID, "material id out of range");
        DevScene S; memset(&S, 0, sizeof(S)); S.materials = mats.data();
        for (size_t i = 0; i < n; ++i) {
            V3 wi(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), wo(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]);
            V3 v = bsdfEval(S, mats[material], wi, wo);
            value3[3 * i] = v.x; value3[3 * i + 1] = v.y; value3[3 * i + 2] = v.z;
            pdf[i] = bsdfPdf(S, mats[material], wi, wo);
        }
        return PHIP_OK;
    } catch (const std::exception &e) { return setErr(PHIP_ERR_INVALID, e.what()); }
}

void phip_debug_host_camera_ray(const phip_camera *cam, const phip_film *film, float sx, float sy, phip_ray *out) {
    DevCamera dc; setupCamera(*cam, *film, dc);
    V3 o, d; float mint, maxt;
    cameraRay(dc, sx, sy, o, d, mint, maxt);
    out->o[0] = o.x; out->o[1] = o.y; out->o[2] = o.z; out->mint = mint;
    out->d[0] = d.x; out->d[1] = d.y; out->d[2] = d.z; out->maxt = maxt;
}

void phip_debug_host_ctr_block(uint32_t pixel, uint32_t sample, uint32_t block, uint32_t seed, float *out4) {
    const U4 h = pcg4d(pixel, sample, block, seed);
    out4[0] = u32ToFloat(h.x); out4[1] = u32ToFloat(h.y); out4[2] = u32ToFloat(h.z); out4[3] = u32ToFloat(h.w);
}

/* DiscreteDistribution::sample (pmf.h:124-136) as the kernels compile it (cdfSample, dv_scene.h: tables of one to three entries are counted, larger
   ones searched): index of each value; `cdf` holds n_entries + 1 values followed by at least three readable floats */
void phip_debug_host_cdf_sample(const float *cdf, uint32_t n_entries, const float *values, size_t n_values, uint32_t *out_index) {
    for (size_t i = 0; i < n_values; ++i) out_index[i] = cdfSample(cdf, n_entries, values[i]);
}

/* PHIP_SAMPLER_LD: the point of request `dim` (2D request q: 2 q; 1D request j: 2 j + 1) of sample `sample` of `pixel` -- the code the kernels compile */
void phip_debug_host_ld_point(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed, uint32_t mask, float *out2) {
    ldPoint(pixel, sample, dim, seed, mask, out2[0], out2[1]);
}

/* PHIP_SAMPLER_SOBOL on the host: index of sample `sample[i]` of pixel (px[i], py[i]) (SobolSampler::setSampleIndex) and its number in dimension dim[i], with the
   row loops of sobolseq.h (byte_tables = 0) or through the byte tables phip_render builds from the same direction numbers (byte_tables = 1: the device's path) --
   the code the kernels compile (dv_math.h) */
int phip_debug_host_sobol(const uint32_t *matrices, uint32_t dims, const unsigned long long *vdc, const unsigned long long *vdc_inv, uint32_t log_res, uint32_t scramble,
                          int byte_tables, size_t n, const uint32_t *sample, const uint32_t *px, const uint32_t *py, const uint32_t *dim,
                          unsigned long long *out_index, float *out_value, float *out_pair2x2 /* 4 n: dimensions dim, dim + 1, dim + 2, dim + 3 in one pass */) {
    std::vector<unsigned long long> rows(2 * PHIP_SOBOL_MATRIX_SIZE, 0ull);
    if (log_res > 1u) for (int i = 0; i < PHIP_SOBOL_MATRIX_SIZE; ++i) { rows[i] = vdc[i]; rows[PHIP_SOBOL_MATRIX_SIZE + i] = vdc_inv[i]; }
    std::vector<uint32_t> bt; std::vector<unsigned long long> vb;
    SobolTab T; memset(&T, 0, sizeof(T));
    T.matrices = matrices; T.vdc = (const uint64_t *) rows.data(); T.vdcInv = (const uint64_t *) rows.data() + PHIP_SOBOL_MATRIX_SIZE;
    T.dims = dims; T.logRes = log_res; T.scramble = scramble; T.resolution = (float) (1u << log_res);
    if (byte_tables) {
        buildSobolByteTables(matrices, dims, rows.data(), bt, vb);
        T.matBt = bt.data(); T.vdcBt = (const uint64_t *) vb.data(); T.vdcInvBt = (const uint64_t *) vb.data() + 4u * 256u;
    }
    for (size_t i = 0; i < n; ++i) {
        if (dim[i] + 5u >= dims) return setErr(PHIP_ERR_INVALID, "dimension out of range");      /* (an index above 2^52 reads on into the next dimension, as sobolseq.h does) */
        const uint64_t idx = sobolSampleIndex(T, sample[i], px[i], py[i]);
        out_index[i] = idx; out_value[i] = sobolSample(T, idx, dim[i]);
        if (out_pair2x2) sobolSample2x2(T, idx, dim[i], dim[i] + 2u, out_pair2x2 + 4 * i);
    }
    return PHIP_OK;
}

/* the radical inverses of PHIP_SAMPLER_HALTON / _HAMMERSLEY (dv_math.h: rinvRadicalInverse) on the host: digit by digit (tables = 0: the loops of qmc.cpp) or
   through the multi-digit tables the render uploads (buildRinvTables) -- tests/test_host_parity.py holds the two against each other and against numpy */
int phip_debug_host_rinv(const uint32_t *primes, const uint16_t *perm, uint32_t dims, int tables, size_t n, const unsigned long long *index, const uint32_t *dim, float *out) {
    std::vector<uint32_t> off(dims); size_t total = 0;
    for (uint32_t d = 0; d < dims; ++d) { off[d] = (uint32_t) total; total += primes[d]; }
    RinvTab T; memset(&T, 0, sizeof(T));
    T.primes = primes; T.perm = perm; T.permOffset = off.data(); T.dims = dims;
    std::vector<uint32_t> di, ch, pw; std::vector<float> fc;
    if (tables) { T.tabDims = buildRinvTables(primes, perm, off.data(), dims, di, ch, fc, pw); T.dimInfo = di.data(); T.chunk = ch.data(); T.fac = fc.data(); T.pw = pw.data(); }
    for (size_t i = 0; i < n; ++i) {
        if (dim[i] >= dims) return setErr(PHIP_ERR_INVALID, "dimension out of range");
        out[i] = rinvRadicalInverse(T, dim[i], index[i]);
    }
    return PHIP_OK;
}

/* ... and the interleaved form of a 2D request (rinvSample2: dimensions dim, dim + 1 in one pass; hammersley = 1: dimension d is the radical inverse in prime d - 1) */
int phip_debug_host_rinv2(const uint32_t *primes, const uint16_t *perm, uint32_t dims, int hammersley, size_t n, const unsigned long long *index, const uint32_t *dim, float *out2) {
    std::vector<uint32_t> off(dims); size_t total = 0;
    for (uint32_t d = 0; d < dims; ++d) { off[d] = (uint32_t) total; total += primes[d]; }
    RinvTab T; memset(&T, 0, sizeof(T));
    T.primes = primes; T.perm = perm; T.permOffset = off.data(); T.dims = dims; T.hammersley = hammersley ? 1u : 0u; T.factor = 1.0f;
    std::vector<uint32_t> di, ch, pw; std::vector<float> fc;
    T.tabDims = buildRinvTables(primes, perm, off.data(), dims, di, ch, fc, pw); T.dimInfo = di.data(); T.chunk = ch.data(); T.fac = fc.data(); T.pw = pw.data();
    for (size_t i = 0; i < n; ++i) {
        if (dim[i] < 1u || dim[i] + 1u >= dims) return setErr(PHIP_ERR_INVALID, "dimension out of range");
        rinvSample2(T, index[i], dim[i], out2[2 * i], out2[2 * i + 1]);
    }
    return PHIP_OK;
}

/* Wald records + BVH statistics of a triangle soup, built exactly like phip_scene_create does (no GPU needed) */
int phip_debug_host_build_bvh(const float *positions, uint32_t n_vertices, const uint32_t *indices, uint32_t n_triangles,
                              phip_accel_info *info, float *scene_box6) {
    try {
        for (uint32_t i = 0; i < 3 * n_triangles; ++i) if (indices[i] >= n_vertices) return setErr(PHIP_ERR_INVALID, "index out of range");
        HostBVH bvh; buildBVH(positions, indices, n_triangles, bvh);
        if (info) {
            info->n_nodes = bvh.nNodes; info->n_leaves = bvh.nLeaves; info->n_triangle_refs = bvh.nTriRefs; info->max_depth = bvh.maxDepth;
            info->node_bytes = 128; info->triangle_bytes = 48; info->sah_cost = bvh.sahCost; info->build_ms = bvh.buildMs;
        }
        if (scene_box6) for (int a = 0; a < 3; ++a) { scene_box6[a] = bvh.sceneMin[a]; scene_box6[3 + a] = bvh.sceneMax[a]; }
        return PHIP_OK;
    } catch (const std::exception &e) { return setErr(PHIP_ERR_INVALID, e.what()); }
}

/* Closest-hit ray casts through the compressed wide BVH ON THE HOST: the tree of buildWide() walked with the node-step
   arithmetic of k_wide.h (wideNodeHits is __host__ __device__) and the Wald test of dv_scene.h.  Lets the CPU test-suite pin
   the builder's encoding (quantisation, slots, child / triangle indexing) and the group stack logic against the oracle
   without a GPU; the device kernels run the same functions.  use_wide = 0 walks nothing and tests every record (brute force). */
int phip_debug_host_trace_wide(const float *positions, uint32_t n_vertices, const uint32_t *indices, uint32_t n_triangles,
                               const phip_ray *rays, size_t n_rays, phip_hit *hits, int use_wide, phip_accel_info *info,
                               uint8_t *seq, uint32_t seq_stride /* optional: the ray's steps in order, 1 = node step, 2 = triangle test, 0 = end */) {
    try {
        for (uint32_t i = 0; i < 3 * n_triangles; ++i) if (indices[i] >= n_vertices) return setErr(PHIP_ERR_INVALID, "index out of range");
        HostBVH bvh; buildBVH(positions, indices, n_triangles, bvh);
        if (info) {
            memset(info, 0, sizeof(*info));
            info->n_nodes = bvh.nWNodes; info->max_depth = bvh.wMaxDepth; info->n_triangle_refs = bvh.nWTris; info->node_bytes = 80; info->triangle_bytes = 48;
            info->sah_cost = bvh.wSahCost; info->build_ms = bvh.buildMs;
        }
        if (use_wide && bvh.nWNodes == 0) return setErr(PHIP_ERR_INVALID, "the scene has no wide tree (a single leaf)");
        DevScene S; memset(&S, 0, sizeof(S));
        for (int a = 0; a < 3; ++a) { S.sceneMin[a] = bvh.sceneMin[a]; S.sceneMax[a] = bvh.sceneMax[a]; }
        const float4 *recs = (const float4 *) (use_wide ? bvh.wtris.data() : bvh.tris.data());
        const size_t nRecs = (use_wide ? bvh.wtris.size() : bvh.tris.size()) / 12;
        const uint4 *wn = (const uint4 *) bvh.wnodes.data();
        for (size_t i = 0; i < n_rays; ++i) {
            const V3 o(rays[i].o[0], rays[i].o[1], rays[i].o[2]), d(rays[i].d[0], rays[i].d[1], rays[i].d[2]);
            phip_hit h; h.t = INFINITY; h.u = h.v = 0; h.prim = PHIP_NO_HIT;
            /* scene-box clip + adaptive epsilon, skdtree.cpp:112-142 (the host twin of clipToScene<false>) */
            float nearT = -INFINITY, farT = INFINITY; bool ok = true;
            const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
            for (int a = 0; a < 3 && ok; ++a) {
                if (dd[a] == 0) { if (oo[a] < S.sceneMin[a] || oo[a] > S.sceneMax[a]) ok = false; }
                else {
                    const float rcp = 1.0f / dd[a];
                    float t1 = (S.sceneMin[a] - oo[a]) * rcp, t2 = (S.sceneMax[a] - oo[a]) * rcp;
                    if (t1 > t2) std::swap(t1, t2);
                    nearT = smax(t1, nearT); farT = smin(t2, farT);
                    if (!(nearT <= farT)) ok = false;
                }
            }
            float mint = nearT, maxt = farT;
            if (ok) {
                float rayMinT = rays[i].mint;
                if (rayMinT == PT_EPSILON) rayMinT *= smax(smax(smax(fabsf(o.x), fabsf(o.y)), fabsf(o.z)), PT_EPSILON);
                if (rayMinT > mint) mint = rayMinT;
                if (rays[i].maxt < maxt) maxt = rays[i].maxt;
                ok = maxt > mint;
            }
            if (ok && !use_wide) {
                for (size_t k = 0; k < nRecs; ++k) {
                    float tu, tv, tt;
                    if (waldIntersect(recs[3 * k], recs[3 * k + 1], recs[3 * k + 2], o, d, mint, maxt, tu, tv, tt) && winsTie(tt, pm_to_bits(recs[3 * k + 2].z), h.t, h.prim)) { maxt = tt; h.t = tt; h.u = tu; h.v = tv; h.prim = pm_to_bits(recs[3 * k + 2].z); }
                }
            } else if (ok) {
                WideRay ray; wideRaySetup(ray, o, d, V3(slabRcp(d.x), slabRcp(d.y), slabRcp(d.z)), mint, maxt);
                std::vector<uint2> stack;
                uint32_t nSeq = 0;
                auto note = [&](uint8_t what) { if (seq && nSeq + 1 < seq_stride) seq[i * seq_stride + nSeq++] = what; };
                uint2 ng = make_uint2(0u, 0x80000000u), tg = make_uint2(0u, 0u);
                for (;;) {
                    if (tg.y == 0u && (ng.y & 0xff000000u)) {
                        const uint32_t bit = 31u - (uint32_t) __builtin_clz(ng.y);
                        ng.y &= ~(1u << bit);
                        if (ng.y & 0xff000000u) stack.push_back(ng);
                        const uint32_t slot = (bit - 24u) ^ (ray.octinv4 & 7u);
                        const uint32_t idx = ng.x + (uint32_t) __builtin_popcount(ng.y & ((1u << slot) - 1u) & 0xffu);
                        if (idx >= bvh.nWNodes) throw std::runtime_error("wide BVH: child index out of range");
                        const uint4 *g = wn + 5 * (size_t) idx;
                        note(1);
                        const uint32_t hitsMask = wideNodeHits(g[0], g[1], g[2], g[3], g[4], ray);
                        ng = make_uint2(g[1].x, (hitsMask & 0xff000000u) | (g[0].w >> 24));
                        tg = make_uint2(g[1].y, hitsMask & 0x00ffffffu);
                    }
                    if (tg.y) {
                        const uint32_t bit = (uint32_t) __builtin_ctz(tg.y);
                        tg.y &= tg.y - 1u;
                        const size_t k = (size_t) tg.x + bit;
                        if (k >= nRecs) throw std::runtime_error("wide BVH: triangle index out of range");
                        note(2);
                        float tu, tv, tt;
                        if (waldIntersect(recs[3 * k], recs[3 * k + 1], recs[3 * k + 2], o, d, ray.mint, ray.maxt, tu, tv, tt) && winsTie(tt, pm_to_bits(recs[3 * k + 2].z), h.t, h.prim)) { ray.maxt = tt; h.t = tt; h.u = tu; h.v = tv; h.prim = pm_to_bits(recs[3 * k + 2].z); }
                    }
                    if (tg.y == 0u && !(ng.y & 0xff000000u)) {
                        if (stack.empty()) break;
                        ng = stack.back(); stack.pop_back();
                    }
                }
            }
            hits[i] = h;
        }
        return PHIP_OK;
    } catch (const std::exception &e) { return setErr(PHIP_ERR_INVALID, e.what()); }
}

} // extern "C"

/* device execution of the phip_fmath.h functions (bitwise host/device agreement test) */
__global__ void k_debug_fmath(int op, size_t n, const float *a, const float *b, float *out) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    switch (op) {
        case 0: pm_sincosf(a[i], &s, &c); out[i] = s; break;
        case 1: pm_sincosf(a[i], &s, &c); out[i] = c; break;
        case 2: out[i] = pm_expf(a[i]); break;
        case 3: out[i] = pm_logf(a[i]); break;
        case 4: out[i] = pm_acosf(a[i]); break;
        case 5: out[i] = pm_atan2f(a[i], b[i]); break;
        case 6: out[i] = pm_tanf(a[i]); break;
        case 7: out[i] = pm_powf(a[i], b[i]); break;
        case 8: out[i] = mts_erf(a[i]); break;
        case 9: out[i] = mts_erfinv(a[i]); break;
        case 10: out[i] = pm_atanf(a[i]); break;
        default: out[i] = 0;
    }
}

/* MIPMap::eval(uv, d0, d1) of dv_scene.h on the host: texture = one phip_texture (levels as delivered) */
extern "C" int phip_debug_host_mip_eval(const phip_texture *t, size_t n, const float *uv2, const float *d0, const float *d1, float *out3) {
    try {
        DevMipLevels lv; memset(&lv, 0, sizeof(lv));
        std::vector<float4> texels;
        lv.nLevels = (int32_t) std::max<uint32_t>(1, t->n_levels);
        int w = (int) t->width, h = (int) t->height;
        for (int l = 0; l < lv.nLevels; ++l) {
            lv.lw[l] = w; lv.lh[l] = h; lv.offset[l] = (uint32_t) texels.size();
            for (size_t i = 0; i < (size_t) w * h; ++i) texels.push_back(make_float4(t->levels[l][3 * i], t->levels[l][3 * i + 1], t->levels[l][3 * i + 2], 0.0f));
            w = std::max(1, (w + 1) / 2); h = std::max(1, (h + 1) / 2);
        }
        lv.bcu = t->wrap_u; lv.bcv = t->wrap_v; lv.filterType = t->filter_type; lv.maxAnisotropy = t->max_anisotropy;
        lv.uvScale[0] = lv.uvScale[1] = 1.0f;
        for (int k = 0; k < 64; ++k) { const float r2 = (float) k / 63.0f; lv.weightLut[k] = pm_expf(-2.0f * r2) - pm_expf(-2.0f); }
        for (size_t i = 0; i < n; ++i) {
            const V3 v = mipEval(texels.data(), lv, V2(uv2[2 * i], uv2[2 * i + 1]), V2(d0[2 * i], d0[2 * i + 1]), V2(d1[2 * i], d1[2 * i + 1]));
            out3[3 * i] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z;
        }
        return 0;
    } catch (const std::exception &e) { return setErr(PHIP_ERR_INVALID, e.what()); }
}

extern "C" int phip_debug_fmath(int on_device, int op, size_t n, const float *a, const float *b, float *out) {
    if (!on_device) {
        for (size_t i = 0; i < n; ++i) {
            float s, c;
            switch (op) {
                case 0: pm_sincosf(a[i], &s, &c); out[i] = s; break;
                case 1: pm_sincosf(a[i], &s, &c); out[i] = c; break;
                case 2: out[i] = pm_expf(a[i]); break;
                case 3: out[i] = pm_logf(a[i]); break;
                case 4: out[i] = pm_acosf(a[i]); break;
                case 5: out[i] = pm_atan2f(a[i], b[i]); break;
                case 6: out[i] = pm_tanf(a[i]); break;
                case 7: out[i] = pm_powf(a[i], b[i]); break;
                case 8: out[i] = mts_erf(a[i]); break;
                case 9: out[i] = mts_erfinv(a[i]); break;
                case 10: out[i] = pm_atanf(a[i]); break;
                default: out[i] = 0;
            }
        }
        return PHIP_OK;
    }
    try {
        DevBuf<float> da, db, dout;
        da.upload(a, n); db.upload(b ? b : a, n); dout.alloc(n);
        hipLaunchKernelGGL(k_debug_fmath, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, op, n, (const float *) da.p, (const float *) db.p, dout.p);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, dout.p, n * sizeof(float), hipMemcpyDeviceToHost));
        return PHIP_OK;
    } catch (const std::exception &e) { return setErr(PHIP_ERR_DEVICE, e.what()); }
}

/* PMC calibration (MI355X_MICROARCH.md, HBM section: FETCH_SIZE is uncalibrated for non-streaming access):
 * n lanes each gather one float4 from a pseudo-random index of a buffer far larger than the 256 MiB
 * Infinity Cache, and n lanes stream-write one float4.  Every gather misses all caches, so the HBM read
 * traffic is n x (line size) and the write traffic n x 16 B: comparing that with FETCH_SIZE / WRITE_SIZE of
 * this kernel gives the correction factors for the traversal kernels' access pattern. */
__global__ void k_debug_gather(const float4 *src, size_t nSrc, float4 *dst, size_t n) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const U4 h = pcg4d((uint32_t) i, (uint32_t) (i >> 32), 12345u, 7u);
    const size_t j = (((size_t) h.x << 32) | h.y) % nSrc;
    dst[i] = src[j];
}
__global__ void k_debug_stream(const float4 *src, float4 *dst, size_t n) {
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

extern "C" int phip_debug_pmc_calibration(size_t src_bytes, size_t n) {
    try {
        DevBuf<float4> src, dst;
        src.alloc(src_bytes / 16); dst.alloc(n);
        HIP_TRY(hipMemset(src.p, 0, src_bytes));
        HIP_TRY(hipMemset(dst.p, 0, n * 16));
        HIP_TRY(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_debug_gather, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, (const float4 *) src.p, src_bytes / 16, dst.p, n);
        hipLaunchKernelGGL(k_debug_stream, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, (const float4 *) src.p, dst.p, n);
        HIP_TRY(hipDeviceSynchronize());
        return PHIP_OK;
    } catch (const std::exception &e) { return setErr(PHIP_ERR_DEVICE, e.what()); }
}

/* ---- the vector-memory roof of a CU under divergence (round 3) ----
 * The ray kernels of the big scenes are bound neither by HBM (7-10 % of the peak) nor by the ALUs (VALU issue 40 %): every lane
 * fetches its own node, so what they load the chip with is LANE-LEVEL 16-byte requests to the CU's texture addresser / L1
 * (TA / TCP).  This micro-benchmark measures what that path sustains, as the ray kernels use it: every lane of a resident grid
 * issues `iters` x 4 independent loads of 16 bytes from pseudo-random places of a buffer that lives in L1 / L2 / Infinity Cache.
 *   mode 0  one random 16-byte element per lane                        (a node / record fetch of k_rays_w)
 *   mode 1  the 4 lanes of a quad read the 4 pieces of one random 64-byte block  (what a transposed, quad-cooperative node fetch would issue)
 *   mode 2  8 lanes share a random 128-byte line        mode 3  16 lanes share 256 bytes
 *   mode 4  as 0 with every second lane switched off    mode 5  as 0 with 16 of the 64 lanes active   (is the cost per instruction or per lane?)
 *   mode 6  five consecutive 16-byte pieces of one random 80-byte record per lane, as five instructions (an 80-byte wide node)
 *   mode 7  one random dword per lane                   mode 8  one random 8 bytes per lane
 *   mode 9  fully coalesced: lane i reads element base + i of a random 1-KB block
 *   mode 10 / 11  as 0 with 8 / 4 of the 64 lanes active
 * Returns the time of the launch; lane_loads = active lanes x loads.  Blocks per CU are pinned with dynamic LDS. */
template <int MODE> __global__ __launch_bounds__(256) void k_debug_vmem(const uint4 *src, uint32_t mask /* elements - 1 */, int iters, uint4 *out) {
    extern __shared__ unsigned char dummy[];
    const uint32_t lane = __lane_id(), gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t grp = MODE == 1 ? 4u : (MODE == 2 ? 8u : (MODE == 3 ? 16u : (MODE == 9 ? 64u : 1u)));
    uint32_t x = (gid / grp) * 2654435761u + 12345u;            /* the lanes of a group draw the same numbers */
    const bool on = MODE == 4 ? (lane & 1u) == 0u : (MODE == 5 ? (lane & 3u) == 0u : (MODE == 10 ? (lane & 7u) == 0u : (MODE == 11 ? (lane & 15u) == 0u : true)));
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (on) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x = x * 1664525u + 1013904223u;
                const uint32_t r = x >> 4;
                if (MODE == 6) {
                    const uint32_t e = (r % ((mask + 1u) / 5u)) * 5u;
#pragma unroll
                    for (int k = 0; k < 5; ++k) { const uint4 v = src[e + k]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
                } else if (MODE == 7) {
                    acc.x ^= ((const uint32_t *) src)[r & (4u * mask + 3u)];
                } else if (MODE == 8) {
                    const uint2 v = ((const uint2 *) src)[r & (2u * mask + 1u)]; acc.x ^= v.x; acc.y ^= v.y;
                } else {
                    const uint32_t e = grp == 1u ? (r & mask) : (((r * grp) & mask) + (lane & (grp - 1u)));
                    const uint4 v = src[e]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
                }
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[gid] = acc;       /* (practically never: keeps the loads alive) */
    if (dummy[0] == 77 && acc.z == 1u) out[0] = acc;
}

extern "C" int phip_debug_vmem_roof(int mode, size_t bytes, int blocks_per_cu, int iters, double *out_ms, double *out_lane_loads) {
    try {
        size_t n = 1; while (n * 2 * 16 <= bytes) n *= 2;       /* 16-byte elements, a power of two */
        DevBuf<uint4> src, out;
        src.alloc(n);
        HIP_TRY(hipMemset(src.p, 0x5a, n * 16));
        int dev = 0, nCU = 256; HIP_TRY(hipGetDevice(&dev));
        { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, dev) == hipSuccess) nCU = prop.multiProcessorCount; }
        const int grid = nCU * blocks_per_cu;
        out.alloc((size_t) grid * 256);
        const size_t lds = (size_t) (160 * 1024 / blocks_per_cu) / 1024 * 1024 - 1024;     /* exactly blocks_per_cu blocks fit a CU */
        typedef void (*K)(const uint4 *, uint32_t, int, uint4 *);
        static const K table[12] = { k_debug_vmem<0>, k_debug_vmem<1>, k_debug_vmem<2>, k_debug_vmem<3>, k_debug_vmem<4>, k_debug_vmem<5>, k_debug_vmem<6>, k_debug_vmem<7>, k_debug_vmem<8>, k_debug_vmem<9>, k_debug_vmem<10>, k_debug_vmem<11> };
        if (mode < 0 || mode > 11) return setErr(PHIP_ERR_INVALID, "mode");
        const K k = table[mode];
        if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, (const uint4 *) src.p, (uint32_t) (n - 1), std::max(1, iters / 8), out.p);     /* warm-up: caches, clocks */
        HIP_TRY(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, (const uint4 *) src.p, (uint32_t) (n - 1), iters, out.p);
        HIP_TRY(hipEventRecord(e1, 0));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
        const double lanesOn = mode == 4 ? 0.5 : (mode == 5 ? 0.25 : (mode == 10 ? 0.125 : (mode == 11 ? 0.0625 : 1.0)));
        if (out_ms) *out_ms = ms;
        if (out_lane_loads) *out_lane_loads = (double) grid * 256.0 * lanesOn * (double) iters * 4.0 * (mode == 6 ? 5.0 : 1.0);
        return PHIP_OK;
    } catch (const std::exception &e) { return setErr(PHIP_ERR_DEVICE, e.what()); }
}
