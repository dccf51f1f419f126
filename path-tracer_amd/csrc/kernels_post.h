// Per-pixel resolve, PPM / display post kernels and the ray-batch loaders of pt_intersect.
// Part of ptgpu.hip (included inside its unnamed namespace, in this order); not a standalone header.

// ------------------------------------------------------------------------------------------------------------
// resolve: per-pixel sum in sample order, then the write of src/trace.rs:166-182
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_resolve(const double *lx, const double *ly, const double *lz, double *ax,
                                                    double *ay, double *az, float *image, uint32_t *frames_per_pixel,
                                                    PassParams pp, uint32_t *counts, unsigned long long *stats,
                                                    uint32_t clear_counts) {
    uint32_t nan_pixels = 0;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < pp.cpix; p += gridDim.x * kBlock) {
        double sx = pp.first_pass ? 0.0 : ax[p];
        double sy = pp.first_pass ? 0.0 : ay[p];
        double sz = pp.first_pass ? 0.0 : az[p];
        for (uint32_t s = 0; s < pp.s_count; ++s) { // pixel_colour += color(...) in sample order
            const uint32_t pid = s * pp.cpix + p;
            sx += lx[pid];
            sy += ly[pid];
            sz += lz[pid];
        }
        if (!pp.last_pass) {
            ax[p] = sx; ay[p] = sy; az[p] = sz;
            continue;
        }
        const uint32_t i = pp.cx0 + p % pp.cw;
        const uint32_t j = pp.cy0 + (p / pp.cw) * pp.row_step;
        const uint64_t pixel = (uint64_t)j * pp.width + i;
        float *px = image + pixel * 4;
        const double ns = (double)pp.spp;
        const double cx = sx / ns, cy = sy / ns, cz = sz / ns; // src/trace.rs:166
        if (pp.mode == PT_MODE_REALTIME) {
            if (pixel_skip_draw(pp.seed, pp.frame, pixel) < 0.8) continue; // src/trace.rs:144-146
            uint32_t c = frames_per_pixel[pixel];
            c += (c <= 1000u) ? 1u : 0u; // src/trace.rs:149
            frames_per_pixel[pixel] = c;
            const float weight = 1.0f / (float)c; // src/trace.rs:171-177
            const float one_minus_weight = 1.0f - weight;
            px[0] = (float)cx * weight + px[0] * one_minus_weight;
            px[1] = (float)cy * weight + px[1] * one_minus_weight;
            px[2] = (float)cz * weight + px[2] * one_minus_weight;
        } else {
            px[0] = (float)cx; // src/trace.rs:179-181
            px[1] = (float)cy;
            px[2] = (float)cz;
        }
        if (px[0] != px[0] || px[1] != px[1] || px[2] != px[2]) nan_pixels += 1;
    }
    // one add per wave: in the scenes whose reference image is mostly NaN, an add per pixel on this one address took
    // longer than the rest of the kernel (a realtime frame of the final scene: 0.40 of 1.49 ms)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nan_pixels += (uint32_t)__shfl_xor((int)nan_pixels, off);
    if ((threadIdx.x & 63u) == 0 && nan_pixels) atomicAdd(&stats[1], (unsigned long long)nan_pixels);
    if (blockIdx.x == 0) { // segments traced in this pass = sum of every bounce's queue fill
        unsigned long long seg = 0;
        const uint32_t total = (pp.max_depth + 1) * kShards;
        for (uint32_t k = threadIdx.x; k < total; k += kBlock) seg += counts[k];
        if (seg) atomicAdd(&stats[0], seg);
        // This block is the last reader of the pass's counters. Where the next pass would otherwise need an operation of
        // its own on the stream to start from zeros (a realtime frame through a pool kernel), the first `clear_counts`
        // words, bounce rows and the chunk counter behind them, are cleared here (each word by the thread that read it).
        for (uint32_t k = threadIdx.x; k < clear_counts; k += kBlock) counts[k] = 0u;
    }
}

// ------------------------------------------------------------------------------------------------------------
// post kernels: PPM conversion and the display transform
// ------------------------------------------------------------------------------------------------------------
// reinhard_tonemap (src/trace.rs:236-242) + sqrt gamma + `as u8` (src/lib.rs:330-340), rows flipped as the PPM
// writer does (src/lib.rs:617-630). f64 with IEEE division and sqrt, as on the host.
__global__ __launch_bounds__(kBlock) void k_tonemap_rgb8(const float *image, uint32_t width, uint32_t height,
                                                         uint8_t *out) {
    const uint32_t n = width * height;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
        const uint32_t i = p % width, j = p / width;
        const float *px = image + ((size_t)(height - j - 1) * width + i) * 4;
        for (int c = 0; c < 3; ++c) {
            const double colour = (double)px[c] * 1.5;
            const double mapped = colour / (colour + 1.0);
            const double scaled = 255.99 * sqrt(mapped);
            uint8_t b; // Rust's `as u8` saturates and maps NaN to 0
            if (!(scaled > 0.0)) b = 0;
            else if (scaled >= 255.0) b = 255;
            else b = (uint8_t)scaled;
            out[(size_t)p * 3 + c] = b;
        }
    }
}
// assets/shaders/tonemap.frag:37-70, f32 like the shader. `v * M` with a column-major mat3 is a dot product of v
// with each constructor triple.
__device__ __forceinline__ float aces_fit(float v) { // RRTAndODTFit, tonemap.frag:37-42
    const float a = v * (v + 0.0245786f) - 0.000090537f;
    const float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}
// What an 8-bit swapchain image keeps of a colour channel in [0, 1]: the nearest of 256 levels (UNORM), after the sRGB
// transfer function when the surface format is an *_SRGB one (the output stage applies it; the shader writes linear values,
// its gamma line is commented out: tonemap.frag:68-69).
__device__ __forceinline__ uint32_t unorm8(float v) { return (uint32_t)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f + 0.5f); }
__device__ __forceinline__ float srgb_oetf(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f; }
// OUT: 0 = f32 RGBA, 1 = RGBA8 UNORM, 2 = RGBA8 with the sRGB transfer function (one 32-bit word per pixel, R in the low byte)
template <int OUT>
__global__ __launch_bounds__(kBlock) void k_display_transform(const float *image, uint32_t width, uint32_t height,
                                                              float exposure, void *out_v) {
    const uint32_t n = width * height;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
        const uint32_t i = p % width, j = p / width; // j = 0 is the top of the screen
        const float *px = image + ((size_t)(height - j - 1) * width + i) * 4;
        const float r = px[0] * exposure, g = px[1] * exposure, b = px[2] * exposure;
        const float ir = r * 0.59719f + g * 0.35458f + b * 0.04823f;
        const float ig = r * 0.07600f + g * 0.90834f + b * 0.01566f;
        const float ib = r * 0.02840f + g * 0.13383f + b * 0.83777f;
        const float fr = aces_fit(ir), fg = aces_fit(ig), fb = aces_fit(ib);
        float orr = fr * 1.60475f + fg * -0.53108f + fb * -0.07367f;
        float org = fr * -0.10208f + fg * 1.10813f + fb * -0.00605f;
        float orb = fr * -0.00327f + fg * -0.07276f + fb * 1.07602f;
        orr = fminf(fmaxf(orr, 0.0f), 1.0f);
        org = fminf(fmaxf(org, 0.0f), 1.0f);
        orb = fminf(fmaxf(orb, 0.0f), 1.0f);
        // uv of the fragment centre after the shader's flip; the distance is the same either way up
        const float u = ((float)i + 0.5f) / (float)width, v = 1.0f - ((float)j + 0.5f) / (float)height;
        const float du = u - 0.5f, dv = v - 0.5f;
        const float vignette = 1.0f - sqrtf(du * du + dv * dv);
        const float cr = orr * vignette, cg = org * vignette, cb = orb * vignette;
        if constexpr (OUT == 0) {
            float *o = (float *)out_v + (size_t)p * 4;
            o[0] = cr;
            o[1] = cg;
            o[2] = cb;
            o[3] = 1.0f;
        } else if constexpr (OUT == 1) {
            ((uint32_t *)out_v)[p] = unorm8(cr) | (unorm8(cg) << 8) | (unorm8(cb) << 16) | 0xff000000u;
        } else {
            ((uint32_t *)out_v)[p] = unorm8(srgb_oetf(cr)) | (unorm8(srgb_oetf(cg)) << 8) | (unorm8(srgb_oetf(cb)) << 16) | 0xff000000u;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// pt_intersect helpers
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_load_rays(const pt_ray *rays, uint32_t n, Queue q) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const pt_ray r = rays[i];
        q.ox(i) = r.origin[0]; q.oy(i) = r.origin[1]; q.oz(i) = r.origin[2];
        q.dx(i) = r.direction[0]; q.dy(i) = r.direction[1]; q.dz(i) = r.direction[2];
        q.tm(i) = r.time;
    }
}
__global__ __launch_bounds__(kBlock) void k_load_rays_soa(const double *soa, uint32_t n, Queue q) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        q.ox(i) = soa[i]; q.oy(i) = soa[(size_t)n + i]; q.oz(i) = soa[2 * (size_t)n + i];
        q.dx(i) = soa[3 * (size_t)n + i]; q.dy(i) = soa[4 * (size_t)n + i]; q.dz(i) = soa[5 * (size_t)n + i];
        q.tm(i) = soa[6 * (size_t)n + i];
    }
}
template <bool X>
__global__ __launch_bounds__(kBlock) void k_store_hits(Queue q, uint32_t n, SceneView sv, const double *hit_t,
                                                       const int32_t *hit_id, pt_hit *out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        pt_hit h;
        const int32_t id = hit_id[i];
        if (id < 0) {
            h.t = 0.0; h.u = 0.0; h.v = 0.0;
            h.p[0] = h.p[1] = h.p[2] = 0.0;
            h.normal[0] = h.normal[1] = h.normal[2] = 0.0;
            h.prim = -1;
            h.material = -1;
        } else {
            const DevLeaf lf = sv.leaves[id & 0x0fffffff];
            const HitRec rec = finalize_hit<X, 3>(sv, lf, d3(q.ox(i), q.oy(i), q.oz(i)),
                                                     d3(q.dx(i), q.dy(i), q.dz(i)), q.tm(i), hit_t[i], (id >> 28) & 7);
            h.t = rec.t; h.u = rec.u; h.v = rec.v;
            h.p[0] = rec.p.x; h.p[1] = rec.p.y; h.p[2] = rec.p.z;
            h.normal[0] = rec.n.x; h.normal[1] = rec.n.y; h.normal[2] = rec.n.z;
            h.prim = rec.prim;
            h.material = rec.mat;
        }
        out[i] = h;
    }
}

// ------------------------------------------------------------------------------------------------------------
// pt_group_gather helpers: the rows y = phase, phase + step, ... of an f32 RGBA image, to and from a packed buffer
// (row k of the buffer = image row phase + k step). One float4 per thread and step.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_pack_rows(const float4 *image, uint32_t width, uint32_t height, uint32_t step,
                                                      uint32_t phase, float4 *packed) {
    const uint64_t n = (uint64_t)group_rows_of(height, step, phase) * width;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const uint32_t k = (uint32_t)(i / width), x = (uint32_t)(i % width);
        packed[i] = image[(uint64_t)group_image_row(step, phase, k) * width + x];
    }
}
// The inverse of k_pack_rows for one member: image row phase + k step = row k of `packed`.
__global__ __launch_bounds__(kBlock) void k_unpack_member_rows(float4 *image, uint32_t width, uint32_t height, uint32_t step,
                                                               uint32_t phase, const float4 *packed) {
    const uint64_t n = (uint64_t)group_rows_of(height, step, phase) * width;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const uint32_t k = (uint32_t)(i / width), x = (uint32_t)(i % width);
        image[(uint64_t)group_image_row(step, phase, k) * width + x] = packed[i];
    }
}
// Every member's packed rows except `skip`'s own (they are in the image already) lie one behind the other in
// `packed`, in rank order; scatters them all into the image.
__global__ __launch_bounds__(kBlock) void k_unpack_rows(float4 *image, uint32_t width, uint32_t height, uint32_t step,
                                                        uint32_t skip, const float4 *packed) {
    const uint64_t n = (uint64_t)height * width;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i % width);
        if (y % step == skip) continue;
        image[i] = packed[group_staged_row(height, step, skip, y) * width + x];
    }
}
