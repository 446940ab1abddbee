/* einx_oracle.c -- CPU ORACLE for the EI-Nexus extract+match hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference algorithm
 * (ZhonghuaYi/EI-Nexus_official, Python/PyTorch) used solely as the checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product path
 * (ei-nexus_official_amd/) never imports, links or calls it.
 *
 * Parity status: PINNED -- every function here is checked against golden vectors captured
 * from the reference itself (tests/golden/gen_golden.py -> tests/golden/<group>.npz, torch 2.10 CPU),
 * see tests/test_oracle_golden.py.
 *
 * Numeric contract (shared with the HIP kernels, include/einx_math.h): fp32 everywhere,
 * dot products are k-ordered fmaf chains starting from +0 (bitwise what v_mfma_f32_32x32x2_f32
 * computes), transcendentals from einx_math.h, reductions in the orders documented per function.
 * File:line citations are relative to the reference repository root.
 */
#include "../include/einx_math.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define EXPORT __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------------------
 * Padder.pad for float tensors: replicate padding (core/modules/utils/util.py:6-32).
 * pads = (w0, w1, h0, h1).
 * ------------------------------------------------------------------------------------- */
EXPORT void orc_pad_replicate(const float* in, int BC, int H, int W, int w0, int w1, int h0, int h1, float* out) {
  const int Hp = H + h0 + h1, Wp = W + w0 + w1;
#pragma omp parallel for
  for (int p = 0; p < BC; ++p)
    for (int y = 0; y < Hp; ++y) {
      int sy = y - h0;
      sy = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy);
      for (int x = 0; x < Wp; ++x) {
        int sx = x - w0;
        sx = sx < 0 ? 0 : (sx > W - 1 ? W - 1 : sx);
        out[((size_t)p * Hp + y) * Wp + x] = in[((size_t)p * H + sy) * W + sx];
      }
    }
}

/* ---------------------------------------------------------------------------------------
 * One conv block: Conv2d(ks in {1,3}, padding ks/2, zero pad) + bias -> [ReLU] ->
 * [BatchNorm2d eval as per-channel affine y = fma(x, scale, shift)] -> [MaxPool2d(2,2)].
 * Block order Conv -> ReLU -> BN is the reference's (core/modules/net/vgg.py:34-38,
 * silk/backbones/superpoint/vgg.py:213-217); SuperPoint has no BN
 * (image_extractors/superpoint_extractor.py:388-406); pooling after the block
 * (core/modules/net/backbone.py:116-123).
 *
 * Canonical accumulation order (the HIP kernel feeds MFMA 32x32x2 in exactly this order):
 *   acc = +0;  for cp in 0..ceil(Cin/2):  for tap in 0..ks*ks:  for h in 0..1:  ci = 2cp+h
 *      acc = fmaf(w[o][ci][tap], x[ci][y+ky-pad][x+kx-pad], acc)      (zero terms are no-ops)
 *   v = acc + bias[o]
 * in: NCHW [B,Cin,H,W]; w: OIHW; out: NCHW [B,Cout,H or H/2,W or W/2].
 * ------------------------------------------------------------------------------------- */
EXPORT void orc_conv_block(const float* in, int B, int Cin, int H, int W, const float* w, const float* bias,
                           const float* scale, const float* shift, int Cout, int ks, int relu, int pool, float* out) {
  const int taps = ks * ks, pad = ks / 2;
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
#pragma omp parallel
  {
    float* acc = (float*)malloc(sizeof(float) * (size_t)W * 2);
    float* row0 = acc;
    float* row1 = acc + W;
#pragma omp for collapse(2) schedule(dynamic, 4)
    for (int b = 0; b < B; ++b)
      for (int o = 0; o < Cout; ++o) {
        const float* wb = w + (size_t)o * Cin * taps;
        const float bo = bias ? bias[o] : 0.0f;
        for (int y = 0; y < H; ++y) {
          float* r = (pool && (y & 1)) ? row1 : row0;
          for (int x = 0; x < W; ++x) r[x] = 0.0f;
          for (int cp = 0; cp < (Cin + 1) / 2; ++cp)
            for (int t = 0; t < taps; ++t) {
              const int ky = t / ks - pad, kx = t % ks - pad;
              const int yy = y + ky;
              if (yy < 0 || yy >= H) continue;
              for (int h = 0; h < 2; ++h) {
                const int ci = 2 * cp + h;
                if (ci >= Cin) continue;
                const float wv = wb[ci * taps + t];
                const float* src = in + (((size_t)b * Cin + ci) * H + yy) * W;
                const int x0 = kx < 0 ? -kx : 0, x1 = kx > 0 ? W - kx : W;
                for (int x = x0; x < x1; ++x) r[x] = fmaf(wv, src[x + kx], r[x]);
              }
            }
          for (int x = 0; x < W; ++x) {
            float v = r[x] + bo;
            if (relu) v = v > 0.0f ? v : 0.0f;
            if (scale) v = fmaf(v, scale[o], shift[o]);
            r[x] = v;
          }
          if (!pool) {
            memcpy(out + (((size_t)b * Cout + o) * Ho + y) * Wo, r, sizeof(float) * W);
          } else if (y & 1) {
            float* dst = out + (((size_t)b * Cout + o) * Ho + (y >> 1)) * Wo;
            for (int x = 0; x < Wo; ++x) {
              const float a = fmaxf(row0[2 * x], row0[2 * x + 1]);
              const float c = fmaxf(row1[2 * x], row1[2 * x + 1]);
              dst[x] = fmaxf(a, c);
            }
          }
        }
      }
    free(acc);
  }
}

/* ---------------------------------------------------------------------------------------
 * logits_to_prob + depth_to_space (core/modules/utils/detector_util.py:18-77).
 * C == 65: softmax over channels (max-subtracted, exp from einx_math, channels summed in
 * order 0..64, p = e / sum), drop dustbin, pixel_shuffle(8): score[8h+i][8w+j] = p[8i+j][h][w].
 * C == 1: p = 1/(1+exp(-x)); score aliases probability.
 * prob: [B,C,hc,wc]; score: [B,hc*cell,wc*cell].
 * ------------------------------------------------------------------------------------- */
EXPORT void orc_logits_to_score(const float* logits, int B, int C, int hc, int wc, float* prob, float* score) {
  if (C == 1) {
    const size_t n = (size_t)B * hc * wc;
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) {
      const float p = einx_sigmoidf(logits[i]);
      prob[i] = p;
      score[i] = p;
    }
    return;
  }
  const int cell = 8, Wp = wc * cell, Hp = hc * cell;
  const size_t plane = (size_t)hc * wc;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < hc; ++h)
      for (int x = 0; x < wc; ++x) {
        const float* l = logits + (size_t)b * C * plane + (size_t)h * wc + x;
        float mx = l[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, l[c * plane]);
        float e[65];
        float s = 0.0f;
        for (int c = 0; c < C; ++c) {
          e[c] = einx_expf(l[c * plane] - mx);
          s = s + e[c];
        }
        for (int c = 0; c < C; ++c) {
          const float p = e[c] / s;
          prob[(size_t)b * C * plane + c * plane + (size_t)h * wc + x] = p;
          if (c < 64) score[((size_t)b * Hp + (h * cell + c / cell)) * Wp + (x * cell + c % cell)] = p;
        }
      }
}

/* ---------------------------------------------------------------------------------------
 * Event-mask handling (core/modules/event_extractors/EventExtractors.py:544-550,561-562):
 * bool mask [B,H,W] is zero-padded to [Hp,Wp], optionally dilated by a 3x3 box (the reference's
 * box-filter > 0), and score is zeroed where the (dilated) mask is false.
 * Then remove_border_points (detector_util.py:138-164) zeroes a `border`-px frame, in place.
 * mask may be NULL (no masking).
 * ------------------------------------------------------------------------------------- */
EXPORT void orc_mask_border(float* score, int B, int Hp, int Wp, const uint8_t* mask, int H, int W, int h0, int w0,
                            int dilate, int border) {
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < Hp; ++y)
      for (int x = 0; x < Wp; ++x) {
        float* s = score + ((size_t)b * Hp + y) * Wp + x;
        if (mask) {
          int on = 0;
          const int r = dilate ? 1 : 0;
          for (int dy = -r; dy <= r && !on; ++dy)
            for (int dx = -r; dx <= r; ++dx) {
              const int yy = y + dy, xx = x + dx; /* padded-map coordinates, zero outside the map */
              if (yy < 0 || yy >= Hp || xx < 0 || xx >= Wp) continue;
              const int sy = yy - h0, sx = xx - w0; /* constant-0 padding of the bool mask */
              if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
              if (mask[((size_t)b * H + sy) * W + sx]) {
                on = 1;
                break;
              }
            }
          if (!on) *s = 0.0f;
        }
        if (border > 0 && (y < border || y >= Hp - border || x < border || x >= Wp - border)) *s = 0.0f;
      }
}

/* ---------------------------------------------------------------------------------------
 * fast_nms (core/modules/utils/detector_util.py:243-337), literal restatement:
 * repeat { is_max = (argmax over the zero-padded (2r+1)^2 window == centre; first maximum wins,
 * so earlier raster taps must be strictly smaller and later taps <= centre);
 * count = #is_max over the WHOLE batch; stop if unchanged; zero every pixel that has a
 * maximum in its window other than itself }.  In place on map [B,H,W].  Returns #iterations.
 * ------------------------------------------------------------------------------------- */
EXPORT int orc_fast_nms(float* map, int B, int H, int W, int r) {
  if (r == 0) return 0;
  const size_t n = (size_t)B * H * W;
  uint8_t* ismax = (uint8_t*)malloc(n);
  long long count = -1;
  int iters = 0;
  for (;;) {
    long long newc = 0;
#pragma omp parallel for collapse(2) reduction(+ : newc)
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          const float* m = map + (size_t)b * H * W;
          const float c = m[(size_t)y * W + x];
          int ok = 1;
          for (int dy = -r; dy <= r && ok; ++dy)
            for (int dx = -r; dx <= r; ++dx) {
              if (dy == 0 && dx == 0) continue;
              const int yy = y + dy, xx = x + dx;
              const float v = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0.0f : m[(size_t)yy * W + xx];
              const int earlier = (dy < 0) || (dy == 0 && dx < 0);
              if (earlier ? !(v < c) : !(v <= c)) {
                ok = 0;
                break;
              }
            }
          ismax[((size_t)b * H + y) * W + x] = (uint8_t)ok;
          newc += ok;
        }
    if (newc == count) break;
    count = newc;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          const uint8_t* im = ismax + (size_t)b * H * W;
          int sup = 0;
          for (int dy = -r; dy <= r && !sup; ++dy)
            for (int dx = -r; dx <= r; ++dx) {
              if (dy == 0 && dx == 0) continue;
              const int yy = y + dy, xx = x + dx;
              if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
              if (im[(size_t)yy * W + xx]) {
                sup = 1;
                break;
              }
            }
          if (sup) map[((size_t)b * H + y) * W + x] = 0.0f;
        }
    ++iters;
  }
  free(ismax);
  return iters;
}

static int cmp_float(const void* a, const void* b) {
  const float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

/* rank indices torch.quantile(q, 'midpoint') gathers, with the reference's fp32 arithmetic:
 * q = float32(N-k)/float32(N) (detector_util.py:112-114), rank = q * float32(N-1)
 * (ATen quantile_compute), lower = floor, upper = ceil. */
EXPORT void orc_topk_ranks(int N, int k, int* lo, int* hi) {
  const float q = (float)(N - k) / (float)N;
  const float rank = q * (float)(N - 1);
  *lo = (int)floorf(rank);
  *hi = (int)ceilf(rank);
}

/* ---------------------------------------------------------------------------------------
 * top-k threshold + thresholding (detector_util.py:108-133): per image
 *   thr_k = 0 if k >= N else lerp(sorted[lo], sorted[hi], 0.5) = b - (b-a)*0.5
 *   thr   = min(thr_k, det_thr)      (top_k == 0 means "no top-k": thr = det_thr)
 *   map   = where(map > thr, map, 0)
 * In place on map [B,N]; writes thr_out[B].
 * ------------------------------------------------------------------------------------- */
EXPORT void orc_topk_threshold(float* map, int B, int N, int top_k, float det_thr, float* thr_out) {
#pragma omp parallel for
  for (int b = 0; b < B; ++b) {
    float* m = map + (size_t)b * N;
    float thr = det_thr;
    if (top_k > 0) {
      float tk;
      if (top_k >= N) {
        tk = 0.0f;
      } else {
        float* s = (float*)malloc(sizeof(float) * N);
        memcpy(s, m, sizeof(float) * N);
        qsort(s, N, sizeof(float), cmp_float);
        int lo, hi;
        orc_topk_ranks(N, top_k, &lo, &hi);
        const float a = s[lo], bb = s[hi];
        tk = bb - (bb - a) * 0.5f;
        free(s);
      }
      thr = fminf(tk, det_thr);
    }
    thr_out[b] = thr;
    for (int i = 0; i < N; ++i)
      if (!(m[i] > thr)) m[i] = 0.0f;
  }
}

/* ---------------------------------------------------------------------------------------
 * prob_map_to_positions_with_prob (detector_util.py:451-484) + Padder.unpad_positions
 * (utils/util.py:52-66) + filter_sparse_feats (EventExtractors.py:496-515):
 * raster-order nonzero of (map > 0); position = index + 0.5 - pad offset; rows whose unpadded
 * coordinate falls outside [0,H)x[0,W) are dropped.  ordering_xy: (x,y,p) instead of (y,x,p).
 * out_pos: [B,cap,3], out_idx: [B,cap] flat padded-map index (for descriptor sampling),
 * counts: [B].
 * ------------------------------------------------------------------------------------- */
EXPORT void orc_positions(const float* map, int B, int Hp, int Wp, int h0, int w0, int H, int W, int ordering_xy, int cap,
                          float* out_pos, int32_t* out_idx, int32_t* counts) {
  for (int b = 0; b < B; ++b) {
    int n = 0;
    for (int y = 0; y < Hp; ++y)
      for (int x = 0; x < Wp; ++x) {
        const float v = map[((size_t)b * Hp + y) * Wp + x];
        if (!(v > 0.0f)) continue;
        const float py = ((float)y + 0.5f) - (float)h0, px = ((float)x + 0.5f) - (float)w0;
        if (!(py >= 0.0f && py < (float)H && px >= 0.0f && px < (float)W)) continue;
        if (n < cap) {
          float* o = out_pos + ((size_t)b * cap + n) * 3;
          o[0] = ordering_xy ? px : py;
          o[1] = ordering_xy ? py : px;
          o[2] = v;
          out_idx[(size_t)b * cap + n] = y * Wp + x;
        }
        ++n;
      }
    counts[b] = n;
  }
}

/* channel-order reduction shared with the HIP kernels: 64 lanes, lane l owns channels
 * l, l+64, ... (sequential fmaf), then xor-butterfly 32,16,8,4,2,1. */
static float lane_butterfly_sumsq(const float* d, int D, int stride) {
  float part[64];
  for (int l = 0; l < 64; ++l) {
    float p = 0.0f;
    for (int c = l; c < D; c += 64) p = fmaf(d[(size_t)c * stride], d[(size_t)c * stride], p);
    part[l] = p;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    float nxt[64];
    for (int l = 0; l < 64; ++l) nxt[l] = part[l] + part[l ^ off];
    memcpy(part, nxt, sizeof(part));
  }
  return part[0];
}

/* ---------------------------------------------------------------------------------------
 * sparsify_low_resolution_descriptors (core/modules/utils/descriptor_util.py:74-128):
 * keypoint at padded-map pixel (y,x): pos = idx+0.5; pos-0.5; 2*(pos/(size-1))-1;
 * grid_sample(bilinear, zeros, align_corners=False): pix = ((g+1)*size_c-1)/2, weights as in
 * ATen's CPU kernel (w = x-floor(x), e = 1-w, ...), value = nw*v_nw + ne*v_ne + sw*v_sw + se*v_se
 * (unfused, in that order); then F.normalize (x / max(||x||,1e-12)) times scale.
 * raw: [B,D,hc,wc]; idx: [B,cap] flat padded index; counts[B]; out: [B,cap,D].
 * ------------------------------------------------------------------------------------- */
EXPORT void orc_desc_sample_bilinear(const float* raw, int B, int D, int hc, int wc, int Hp, int Wp, const int32_t* idx,
                                     const int32_t* counts, int cap, float scale, float* out) {
#pragma omp parallel for
  for (int b = 0; b < B; ++b) {
    float* tmp = (float*)malloc(sizeof(float) * D);
    for (int i = 0; i < counts[b] && i < cap; ++i) {
      const int fi = idx[(size_t)b * cap + i];
      const int y = fi / Wp, x = fi % Wp;
      float py = ((float)y + 0.5f) - 0.5f, px = ((float)x + 0.5f) - 0.5f;
      const float gy = 2.0f * (py / (float)(Hp - 1)) - 1.0f;
      const float gx = 2.0f * (px / (float)(Wp - 1)) - 1.0f;
      const float iy = ((gy + 1.0f) * (float)hc - 1.0f) / 2.0f;
      const float ix = ((gx + 1.0f) * (float)wc - 1.0f) / 2.0f;
      const float fx = floorf(ix), fy = floorf(iy);
      const float w = ix - fx, e = 1.0f - w, n = iy - fy, s = 1.0f - n;
      const float nw = s * e, ne = s * w, sw = n * e, se = n * w;
      const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
      const int vx0 = x0 >= 0 && x0 < wc, vx1 = x1 >= 0 && x1 < wc, vy0 = y0 >= 0 && y0 < hc, vy1 = y1 >= 0 && y1 < hc;
      for (int c = 0; c < D; ++c) {
        const float* p = raw + ((size_t)b * D + c) * hc * wc;
        const float a = (vy0 && vx0) ? p[y0 * wc + x0] : 0.0f;
        const float bb = (vy0 && vx1) ? p[y0 * wc + x1] : 0.0f;
        const float cc = (vy1 && vx0) ? p[y1 * wc + x0] : 0.0f;
        const float dd = (vy1 && vx1) ? p[y1 * wc + x1] : 0.0f;
        float t = a * nw;
        t = t + bb * ne;
        t = t + cc * sw;
        t = t + dd * se;
        tmp[c] = t;
      }
      const float nrm = sqrtf(lane_butterfly_sumsq(tmp, D, 1));
      const float den = fmaxf(nrm, 1e-12f);
      float* o = out + ((size_t)b * cap + i) * D;
      for (int c = 0; c < D; ++c) o[c] = scale * (tmp[c] / den);
    }
    free(tmp);
  }
}

/* sparsify_full_resolution_descriptors (descriptor_util.py:50-71): integer gather at
 * floor(pos) then L2-normalise x scale.  raw: [B,D,H,W]. */
EXPORT void orc_desc_gather(const float* raw, int B, int D, int H, int W, const int32_t* idx, const int32_t* counts, int cap,
                            float scale, float* out) {
#pragma omp parallel for
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < counts[b] && i < cap; ++i) {
      const int fi = idx[(size_t)b * cap + i];
      const float* p = raw + (size_t)b * D * H * W + fi;
      const float nrm = sqrtf(lane_butterfly_sumsq(p, D, H * W));
      const float den = fmaxf(nrm, 1e-12f);
      float* o = out + ((size_t)b * cap + i) * D;
      for (int c = 0; c < D; ++c) o[c] = scale * (p[(size_t)c * H * W] / den);
    }
}

/* normalize_descriptors over dim=1 of a dense map (descriptor_util.py:21-28); per pixel the
 * squared norm is a sequential fmaf chain over channels 0..D-1.  raw/out: [B,D,P]. */
EXPORT void orc_normalize_map(const float* raw, int B, int D, int P, float scale, float* out) {
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int p = 0; p < P; ++p) {
      const float* r = raw + (size_t)b * D * P + p;
      float s = 0.0f;
      for (int c = 0; c < D; ++c) s = fmaf(r[(size_t)c * P], r[(size_t)c * P], s);
      const float den = fmaxf(sqrtf(s), 1e-12f);
      for (int c = 0; c < D; ++c) out[(size_t)b * D * P + (size_t)c * P + p] = scale * (r[(size_t)c * P] / den);
    }
}

/* F.normalize(x, dim=1) * scale on a row-major [R,C] matrix: the random padding descriptors of
 * the un-frozen Matcher branch (core/modules/Matchers.py:114-131); reduction order as the sparse
 * descriptor kernels (lane_butterfly_sumsq). */
EXPORT void orc_normalize_rows(const float* x, int R, int C, float scale, float* out) {
  for (int r = 0; r < R; ++r) {
    const float* v = x + (size_t)r * C;
    const float den = fmaxf(sqrtf(lane_butterfly_sumsq(v, C, 1)), 1e-12f);
    for (int c = 0; c < C; ++c) out[(size_t)r * C + c] = scale * (v[c] / den);
  }
}

/* upsample_descriptors (descriptor_util.py:131-138): bilinear resize (align_corners=False,
 * no antialias: src = max((dst+0.5)*in/out-0.5, 0), upper neighbour clamped) + normalize.
 * raw: [B,D,hc,wc] -> out: [B,D,Ho,Wo].  Interpolation as ATen upsample_bilinear2d:
 * v = w0y*(w0x*a + w1x*b) + w1y*(w0x*c + w1x*d). */
EXPORT void orc_upsample_normalize(const float* raw, int B, int D, int hc, int wc, int Ho, int Wo, float scale, float* out) {
  const float sy = (float)hc / (float)Ho, sx = (float)wc / (float)Wo;
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < Ho; ++y) {
      float* tmp = (float*)malloc(sizeof(float) * D);
      float fy = ((float)y + 0.5f) * sy - 0.5f;
      if (fy < 0.0f) fy = 0.0f;
      const int y0 = (int)fy, y1 = y0 + (y0 < hc - 1 ? 1 : 0);
      const float ly = fy - (float)y0, hy = 1.0f - ly;
      for (int x = 0; x < Wo; ++x) {
        float fx = ((float)x + 0.5f) * sx - 0.5f;
        if (fx < 0.0f) fx = 0.0f;
        const int x0 = (int)fx, x1 = x0 + (x0 < wc - 1 ? 1 : 0);
        const float lx = fx - (float)x0, hx = 1.0f - lx;
        float s = 0.0f;
        for (int c = 0; c < D; ++c) {
          const float* p = raw + ((size_t)b * D + c) * hc * wc;
          const float v = hy * (hx * p[y0 * wc + x0] + lx * p[y0 * wc + x1]) + ly * (hx * p[y1 * wc + x0] + lx * p[y1 * wc + x1]);
          tmp[c] = v;
          s = fmaf(v, v, s);
        }
        const float den = fmaxf(sqrtf(s), 1e-12f);
        for (int c = 0; c < D; ++c) out[(((size_t)b * D + c) * Ho + y) * Wo + x] = scale * (tmp[c] / den);
      }
      free(tmp);
    }
}

/* ---------------------------------------------------------------------------------------
 * NearestNeighborMatcher (core/modules/matchers/MNN.py:43-140) for one pair:
 * sim = d0 d1^T (k-ordered fmaf chain), matches0 = row arg-max, matches1 = column arg-max
 * (first maximum on ties), mutual check (:25-32), scores = (match > -1),
 * log_assignment[:n,:m] = log_softmax(sim,-1) + log_softmax(sim,-2), last row/col 0 (:96-98).
 * la may be NULL.  sim_out may be NULL.
 * ------------------------------------------------------------------------------------- */
EXPORT void orc_mnn(const float* d0, int n, const float* d1, int m, int D, int64_t* m0, int64_t* m1, float* s0, float* s1,
                    float* la, float* sim_out) {
  float* sim = (float*)malloc(sizeof(float) * (size_t)n * m);
#pragma omp parallel for
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < D; ++k) acc = fmaf(d0[(size_t)i * D + k], d1[(size_t)j * D + k], acc);
      sim[(size_t)i * m + j] = acc;
    }
  for (int i = 0; i < n; ++i) {
    int best = 0;
    for (int j = 1; j < m; ++j)
      if (sim[(size_t)i * m + j] > sim[(size_t)i * m + best]) best = j;
    m0[i] = best;
  }
  for (int j = 0; j < m; ++j) {
    int best = 0;
    for (int i = 1; i < n; ++i)
      if (sim[(size_t)i * m + j] > sim[(size_t)best * m + j]) best = i;
    m1[j] = best;
  }
  int64_t* t0 = (int64_t*)malloc(sizeof(int64_t) * n);
  memcpy(t0, m0, sizeof(int64_t) * n);
  for (int i = 0; i < n; ++i)
    if (m1[m0[i]] != i) m0[i] = -1;
  for (int j = 0; j < m; ++j)
    if (t0[m1[j]] != j) m1[j] = -1;
  free(t0);
  for (int i = 0; i < n; ++i) s0[i] = m0[i] > -1 ? 1.0f : 0.0f;
  for (int j = 0; j < m; ++j) s1[j] = m1[j] > -1 ? 1.0f : 0.0f;
  if (la) {
    float* rmax = (float*)malloc(sizeof(float) * n);
    float* rlse = (float*)malloc(sizeof(float) * n);
    float* cmax = (float*)malloc(sizeof(float) * m);
    float* clse = (float*)malloc(sizeof(float) * m);
    for (int i = 0; i < n; ++i) {
      float mx = sim[(size_t)i * m];
      for (int j = 1; j < m; ++j) mx = fmaxf(mx, sim[(size_t)i * m + j]);
      float s = 0.0f;
      for (int j = 0; j < m; ++j) s += einx_expf(sim[(size_t)i * m + j] - mx);
      rmax[i] = mx;
      rlse[i] = einx_logf(s);
    }
    for (int j = 0; j < m; ++j) {
      float mx = sim[j];
      for (int i = 1; i < n; ++i) mx = fmaxf(mx, sim[(size_t)i * m + j]);
      float s = 0.0f;
      for (int i = 0; i < n; ++i) s += einx_expf(sim[(size_t)i * m + j] - mx);
      cmax[j] = mx;
      clse[j] = einx_logf(s);
    }
    for (int i = 0; i <= n; ++i)
      for (int j = 0; j <= m; ++j) {
        float v = 0.0f;
        if (i < n && j < m) {
          const float sv = sim[(size_t)i * m + j];
          v = ((sv - rmax[i]) - rlse[i]) + ((sv - cmax[j]) - clse[j]);
        }
        la[(size_t)i * (m + 1) + j] = v;
      }
    free(rmax);
    free(rlse);
    free(cmax);
    free(clse);
  }
  if (sim_out) memcpy(sim_out, sim, sizeof(float) * (size_t)n * m);
  free(sim);
}

/* find_nn with the optional thresholds (core/modules/matchers/MNN.py:11-22) + mutual_check (:25-32).
 * k = 2 if ratio_thresh else 1; sim.topk(k) per row (per column for matches1, :89-92):
 *   dist = 2 * (1 - sim_nn)                                  (fp32)
 *   mask &= dist[0] <= ratio_thresh**2 * dist[1]             (scalar rounded to fp32 by torch, product in fp32)
 *   mask &= dist[0] <= distance_thresh**2
 * The second neighbour counts multiplicity (topk returns an equal value twice), the first index wins ties.
 * use_ratio needs at least two candidates per row and per column (torch.topk raises otherwise): returns -1. */
static void find_nn_rows(const float* sim, int n, int m, size_t si, size_t sj, int use_ratio, float ratio_sq, int use_dist, float dist_sq,
                         int64_t* out) {
  for (int i = 0; i < n; ++i) {
    int best = 0;
    for (int j = 1; j < m; ++j)
      if (sim[i * si + j * sj] > sim[i * si + best * sj]) best = j;
    const float d0 = 2.0f * (1.0f - sim[i * si + best * sj]);
    int ok = 1;
    if (use_ratio) {
      int sec = -1;
      for (int j = 0; j < m; ++j) {
        if (j == best) continue;
        if (sec < 0 || sim[i * si + j * sj] > sim[i * si + sec * sj]) sec = j;
      }
      const float d1 = 2.0f * (1.0f - sim[i * si + sec * sj]);
      ok = ok && (d0 <= ratio_sq * d1);
    }
    if (use_dist) ok = ok && (d0 <= dist_sq);
    out[i] = ok ? best : -1;
  }
}

EXPORT int orc_mnn_thresh(const float* d0, int n, const float* d1, int m, int D, int use_ratio, float ratio_sq, int use_dist, float dist_sq,
                          int64_t* m0, int64_t* m1, float* s0, float* s1) {
  if (use_ratio && (n < 2 || m < 2)) return -1;
  float* sim = (float*)malloc(sizeof(float) * (size_t)n * m);
#pragma omp parallel for
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < D; ++k) acc = fmaf(d0[(size_t)i * D + k], d1[(size_t)j * D + k], acc);
      sim[(size_t)i * m + j] = acc;
    }
  find_nn_rows(sim, n, m, (size_t)m, 1, use_ratio, ratio_sq, use_dist, dist_sq, m0);
  find_nn_rows(sim, m, n, 1, (size_t)m, use_ratio, ratio_sq, use_dist, dist_sq, m1);
  int64_t* t0 = (int64_t*)malloc(sizeof(int64_t) * n);
  memcpy(t0, m0, sizeof(int64_t) * n);
  for (int i = 0; i < n; ++i)  /* loop0 = m1[m0 > -1 ? m0 : 0]; keep m0 if m0 > -1 and loop0 == i */
    if (!(m0[i] > -1 && m1[m0[i]] == i)) m0[i] = -1;
  for (int j = 0; j < m; ++j)
    if (!(m1[j] > -1 && t0[m1[j]] == j)) m1[j] = -1;
  free(t0);
  for (int i = 0; i < n; ++i) s0[i] = m0[i] > -1 ? 1.0f : 0.0f;
  for (int j = 0; j < m; ++j) s1[j] = m1[j] > -1 ? 1.0f : 0.0f;
  free(sim);
  return 0;
}

/* =======================================================================================
 * LightGlue building blocks (core/modules/matchers/lightglue.py).
 * ===================================================================================== */

/* Linear: y[r][o] = (sum_k x[r][k] w[o][k], k-ordered fmaf chain from +0) + b[o]. */
EXPORT void orc_linear(const float* x, int R, int K, const float* w, const float* b, int O, float* y) {
#pragma omp parallel for
  for (int r = 0; r < R; ++r)
    for (int o = 0; o < O; ++o) {
      float acc = 0.0f;
      for (int k = 0; k < K; ++k) acc = fmaf(x[(size_t)r * K + k], w[(size_t)o * K + k], acc);
      y[(size_t)r * O + o] = b ? acc + b[o] : acc;
    }
}

/* normalize_keypoints (:137-148) + LearnableFourierPositionalEncoding (:161-174).
 * kpts: [n,2] (first two columns of sparse_positions, stride `kstride`), size = (s0,s1).
 * enc: [2][n][dh]: cos / sin of Wr k, each value repeated twice along the last dim; dh = head dim
 * (descriptor_dim // num_heads, lightglue.py:456-459), Wr: [dh/2][2]. */
EXPORT void orc_lg_posenc_dh(const float* kpts, int kstride, int n, float s0, float s1, const float* Wr, int dh, float* enc) {
  const float sh0 = s0 / 2.0f, sh1 = s1 / 2.0f;
  const float sc = fmaxf(s0, s1) / 2.0f;
  for (int i = 0; i < n; ++i) {
    const float k0 = (kpts[(size_t)i * kstride + 0] - sh0) / sc;
    const float k1 = (kpts[(size_t)i * kstride + 1] - sh1) / sc;
    for (int f = 0; f < dh / 2; ++f) {
      float p = fmaf(k0, Wr[f * 2 + 0], 0.0f);
      p = fmaf(k1, Wr[f * 2 + 1], p);
      float sn, cs;
      einx_sincosf(p, &sn, &cs);
      enc[((size_t)0 * n + i) * dh + 2 * f] = cs;
      enc[((size_t)0 * n + i) * dh + 2 * f + 1] = cs;
      enc[((size_t)1 * n + i) * dh + 2 * f] = sn;
      enc[((size_t)1 * n + i) * dh + 2 * f + 1] = sn;
    }
  }
}

/* 64-wide heads (the LightGlue default: 256 = 4 x 64) */
EXPORT void orc_lg_posenc(const float* kpts, int kstride, int n, float s0, float s1, const float* Wr /*[32][2]*/, float* enc) {
  orc_lg_posenc_dh(kpts, kstride, n, s0, s1, Wr, 64, enc);
}

/* softmax(q k^T * scale) v for one head; q:[n,dh] k,v:[m,dh] with row strides; out [n,dh] stride so.
 * scores: k-ordered fmaf chain, times scale; softmax max-subtracted, exps summed j ascending,
 * PV: acc_j ascending fmaf(p_j, v_j), p_j = e_j / sum.  (F.scaled_dot_product_attention,
 * lightglue.py:226-229; einsum+softmax+einsum for cross attention :317-324.) */
static void attn_head(const float* q, int sq, const float* k, int sk, const float* v, int sv, int n, int m, int dh, float scale,
                      float* out, int so) {
#pragma omp parallel for
  for (int i = 0; i < n; ++i) {
    float* s = (float*)malloc(sizeof(float) * m);
    float mx = -INFINITY;
    for (int j = 0; j < m; ++j) {
      float acc = 0.0f;
      for (int d = 0; d < dh; ++d) acc = fmaf(q[(size_t)i * sq + d], k[(size_t)j * sk + d], acc);
      acc = acc * scale;
      s[j] = acc;
      mx = fmaxf(mx, acc);
    }
    float sum = 0.0f;
    for (int j = 0; j < m; ++j) {
      s[j] = einx_expf(s[j] - mx);
      sum += s[j];
    }
    for (int d = 0; d < dh; ++d) {
      float acc = 0.0f;
      for (int j = 0; j < m; ++j) acc = fmaf(s[j] / sum, v[(size_t)j * sv + d], acc);
      out[(size_t)i * so + d] = acc;
    }
    free(s);
  }
}

/* ffn(cat[x,msg]) + residual (:250-256,272): Linear(2d,2d) -> LayerNorm(2d, eps 1e-5) -> GELU ->
 * Linear(2d,d); x += result.  LayerNorm: mean and biased variance as sequential sums over the
 * 2d features (two-pass), y = (v-mean)/sqrt(var+eps)*g + b. */
static void ffn_residual(float* x, const float* msg, int n, int d, const float* w0, const float* b0, const float* g,
                         const float* be, const float* w3, const float* b3) {
  const int d2 = 2 * d;
#pragma omp parallel for
  for (int r = 0; r < n; ++r) {
    float* cat = (float*)malloc(sizeof(float) * d2 * 2);
    float* h = cat + d2;
    memcpy(cat, x + (size_t)r * d, sizeof(float) * d);
    memcpy(cat + d, msg + (size_t)r * d, sizeof(float) * d);
    for (int o = 0; o < d2; ++o) {
      float acc = 0.0f;
      for (int k = 0; k < d2; ++k) acc = fmaf(cat[k], w0[(size_t)o * d2 + k], acc);
      h[o] = acc + b0[o];
    }
    float mean = 0.0f;
    for (int o = 0; o < d2; ++o) mean += h[o];
    mean = mean / (float)d2;
    float var = 0.0f;
    for (int o = 0; o < d2; ++o) var = fmaf(h[o] - mean, h[o] - mean, var);
    var = var / (float)d2;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    for (int o = 0; o < d2; ++o) h[o] = einx_geluf(fmaf((h[o] - mean) * rstd, g[o], be[o]));
    for (int o = 0; o < d; ++o) {
      float acc = 0.0f;
      for (int k = 0; k < d2; ++k) acc = fmaf(h[k], w3[(size_t)o * d2 + k], acc);
      x[(size_t)r * d + o] = x[(size_t)r * d + o] + (acc + b3[o]);
    }
    free(cat);
  }
}

/* SelfBlock.forward (:258-272). x:[n,d] in place; enc:[2][n][dh]; d = heads*dh. */
EXPORT void orc_lg_self_block(float* x, int n, int d, int heads, const float* enc, const float* Wqkv, const float* bqkv,
                              const float* Wo, const float* bo, const float* w0, const float* b0, const float* g,
                              const float* be, const float* w3, const float* b3) {
  const int dh = d / heads;
  float* qkv = (float*)malloc(sizeof(float) * (size_t)n * 3 * d);
  orc_linear(x, n, d, Wqkv, bqkv, 3 * d, qkv);
  /* unflatten(-1,(heads,dh,3)): feature index = (h*dh + c)*3 + t */
  float* q = (float*)malloc(sizeof(float) * (size_t)n * d * 3);
  float* k = q + (size_t)n * d;
  float* v = k + (size_t)n * d;
  for (int i = 0; i < n; ++i)
    for (int h = 0; h < heads; ++h)
      for (int c = 0; c < dh; ++c) {
        const float* src = qkv + (size_t)i * 3 * d + ((size_t)h * dh + c) * 3;
        q[(size_t)i * d + h * dh + c] = src[0];
        k[(size_t)i * d + h * dh + c] = src[1];
        v[(size_t)i * d + h * dh + c] = src[2];
      }
  /* rotary (:151-158): t*cos + rotate_half(t)*sin, rotate_half pairs (x0,x1)->(-x1,x0) */
  for (int t = 0; t < 2; ++t) {
    float* a = t == 0 ? q : k;
    for (int i = 0; i < n; ++i)
      for (int h = 0; h < heads; ++h)
        for (int c = 0; c < dh; c += 2) {
          float* p = a + (size_t)i * d + h * dh + c;
          const float x0 = p[0], x1 = p[1];
          const float c0 = enc[((size_t)0 * n + i) * dh + c], c1 = enc[((size_t)0 * n + i) * dh + c + 1];
          const float s0 = enc[((size_t)1 * n + i) * dh + c], s1 = enc[((size_t)1 * n + i) * dh + c + 1];
          p[0] = (x0 * c0) + ((-x1) * s0);
          p[1] = (x1 * c1) + (x0 * s1);
        }
  }
  float* ctx = (float*)malloc(sizeof(float) * (size_t)n * d);
  const float scale = 1.0f / sqrtf((float)dh);
  for (int h = 0; h < heads; ++h) attn_head(q + h * dh, d, k + h * dh, d, v + h * dh, d, n, n, dh, scale, ctx + h * dh, d);
  float* msg = (float*)malloc(sizeof(float) * (size_t)n * d);
  orc_linear(ctx, n, d, Wo, bo, d, msg);
  ffn_residual(x, msg, n, d, w0, b0, g, be, w3, b3);
  free(qkv);
  free(q);
  free(ctx);
  free(msg);
}

/* CrossBlock.forward (:303-330), non-flash branch: qk scaled by dh^-1/4 on both sides, one sim,
 * softmax over j for m0 and over i for m1. */
EXPORT void orc_lg_cross_block(float* x0, int n, float* x1, int m, int d, int heads, const float* Wqk, const float* bqk,
                               const float* Wv, const float* bv, const float* Wo, const float* bo, const float* w0,
                               const float* b0, const float* g, const float* be, const float* w3, const float* b3) {
  const int dh = d / heads;
  float* qk0 = (float*)malloc(sizeof(float) * (size_t)(n + m) * d * 4);
  float* qk1 = qk0 + (size_t)n * d;
  float* v0 = qk1 + (size_t)m * d;
  float* v1 = v0 + (size_t)n * d;
  float* c0 = v1 + (size_t)m * d;
  float* c1 = c0 + (size_t)n * d;
  float* m0 = c1 + (size_t)m * d;
  float* m1 = m0 + (size_t)n * d;
  orc_linear(x0, n, d, Wqk, bqk, d, qk0);
  orc_linear(x1, m, d, Wqk, bqk, d, qk1);
  orc_linear(x0, n, d, Wv, bv, d, v0);
  orc_linear(x1, m, d, Wv, bv, d, v1);
  const float s = sqrtf(1.0f / sqrtf((float)dh)); /* scale**0.5 with scale = dh**-0.5 */
  for (size_t i = 0; i < (size_t)n * d; ++i) qk0[i] = qk0[i] * s;
  for (size_t i = 0; i < (size_t)m * d; ++i) qk1[i] = qk1[i] * s;
  for (int h = 0; h < heads; ++h) {
    attn_head(qk0 + h * dh, d, qk1 + h * dh, d, v1 + h * dh, d, n, m, dh, 1.0f, c0 + h * dh, d);
    attn_head(qk1 + h * dh, d, qk0 + h * dh, d, v0 + h * dh, d, m, n, dh, 1.0f, c1 + h * dh, d);
  }
  orc_linear(c0, n, d, Wo, bo, d, m0);
  orc_linear(c1, m, d, Wo, bo, d, m1);
  ffn_residual(x0, m0, n, d, w0, b0, g, be, w3, b3);
  ffn_residual(x1, m1, m, d, w0, b0, g, be, w3, b3);
  free(qk0);
}

/* MatchAssignment.forward + sigmoid_log_double_softmax + filter_matches (:365-418).
 * scores [n+1,m+1]; matches/mscores per side. */
EXPORT void orc_lg_assign(const float* x0, int n, const float* x1, int m, int d, const float* Wp, const float* bp,
                          const float* wm, const float* bm, float th, float* scores, int64_t* m0, int64_t* m1, float* ms0,
                          float* ms1) {
  float* md0 = (float*)malloc(sizeof(float) * (size_t)(n + m) * d);
  float* md1 = md0 + (size_t)n * d;
  orc_linear(x0, n, d, Wp, bp, d, md0);
  orc_linear(x1, m, d, Wp, bp, d, md1);
  const float div = sqrtf(sqrtf((float)d)); /* d**0.25 */
  for (size_t i = 0; i < (size_t)(n + m) * d; ++i) md0[i] = md0[i] / div;
  float* z0 = (float*)malloc(sizeof(float) * (n + m));
  float* z1 = z0 + n;
  orc_linear(x0, n, d, wm, bm, 1, z0);
  orc_linear(x1, m, d, wm, bm, 1, z1);
  float* sim = (float*)malloc(sizeof(float) * (size_t)n * m);
#pragma omp parallel for
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < d; ++k) acc = fmaf(md0[(size_t)i * d + k], md1[(size_t)j * d + k], acc);
      sim[(size_t)i * m + j] = acc;
    }
  float* rmax = (float*)malloc(sizeof(float) * 2 * (n + m));
  float* rlse = rmax + n;
  float* cmax = rlse + n;
  float* clse = cmax + m;
  for (int i = 0; i < n; ++i) {
    float mx = -INFINITY, sm = 0.0f;
    for (int j = 0; j < m; ++j) mx = fmaxf(mx, sim[(size_t)i * m + j]);
    for (int j = 0; j < m; ++j) sm += einx_expf(sim[(size_t)i * m + j] - mx);
    rmax[i] = mx;
    rlse[i] = einx_logf(sm);
  }
  for (int j = 0; j < m; ++j) {
    float mx = -INFINITY, sm = 0.0f;
    for (int i = 0; i < n; ++i) mx = fmaxf(mx, sim[(size_t)i * m + j]);
    for (int i = 0; i < n; ++i) sm += einx_expf(sim[(size_t)i * m + j] - mx);
    cmax[j] = mx;
    clse[j] = einx_logf(sm);
  }
  const int M1 = m + 1;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const float sv = sim[(size_t)i * m + j];
      const float cert = einx_logsigmoidf(z0[i]) + einx_logsigmoidf(z1[j]);
      scores[(size_t)i * M1 + j] = (((sv - rmax[i]) - rlse[i]) + ((sv - cmax[j]) - clse[j])) + cert;
    }
  for (int i = 0; i < n; ++i) scores[(size_t)i * M1 + m] = einx_logsigmoidf(-z0[i]);
  for (int j = 0; j < m; ++j) scores[(size_t)n * M1 + j] = einx_logsigmoidf(-z1[j]);
  scores[(size_t)n * M1 + m] = 0.0f;
  /* filter_matches */
  float* max0 = (float*)malloc(sizeof(float) * n);
  for (int i = 0; i < n; ++i) {
    int best = 0;
    for (int j = 1; j < m; ++j)
      if (scores[(size_t)i * M1 + j] > scores[(size_t)i * M1 + best]) best = j;
    m0[i] = best;
    max0[i] = scores[(size_t)i * M1 + best];
  }
  for (int j = 0; j < m; ++j) {
    int best = 0;
    for (int i = 1; i < n; ++i)
      if (scores[(size_t)i * M1 + j] > scores[(size_t)best * M1 + j]) best = i;
    m1[j] = best;
  }
  for (int i = 0; i < n; ++i) {
    const int mutual = (m1[m0[i]] == i);
    ms0[i] = mutual ? einx_expf(max0[i]) : 0.0f;
  }
  int64_t* t0 = (int64_t*)malloc(sizeof(int64_t) * n);
  memcpy(t0, m0, sizeof(int64_t) * n);
  for (int j = 0; j < m; ++j) {
    const int mutual = (t0[m1[j]] == j);
    ms1[j] = mutual ? ms0[m1[j]] : 0.0f;
  }
  for (int i = 0; i < n; ++i) {
    const int mutual = (m1[t0[i]] == i);
    if (!(mutual && ms0[i] > th)) m0[i] = -1;
  }
  for (int j = 0; j < m; ++j) {
    const int64_t i = m1[j];
    const int mutual = (t0[i] == j);
    if (!(mutual && m0[i] > -1)) m1[j] = -1;
  }
  free(t0);
  free(max0);
  free(rmax);
  free(sim);
  free(z0);
  free(md0);
}

/* =======================================================================================
 * Event representation (the step before the path).
 * ===================================================================================== */

/* events_to_voxel_grid (datasets/representations.py:67-124) incl. time_normalization (:8-21):
 * float64 time normalisation, fp32 trilinear weights, sequential accumulation in the reference's
 * loop order (corner loops outermost, events in order), then (v-mean)/std(unbiased) on non-zeros. */
EXPORT void orc_voxel_grid(const float* x, const float* y, const double* t, const float* p, long long n, int bins, int H, int W,
                           int normalize, float* grid) {
  const size_t per = (size_t)bins * H * W;
  memset(grid, 0, per * sizeof(float));
  if (n <= 0) return;
  const double t0d = t[0], den = (t[n - 1] - t[0]) + 1e-8;
  const float tf0 = (float)(0.0 / den), tfl = (float)((t[n - 1] - t0d) / den);
  for (int dx = 0; dx < 2; ++dx)
    for (int dy = 0; dy < 2; ++dy)
      for (int dt = 0; dt < 2; ++dt)
        for (long long i = 0; i < n; ++i) {
          const float tf = (float)((t[i] - t0d) / den);
          const float tn = ((float)(bins - 1) * (tf - tf0)) / (tfl - tf0);
          float value = p[i];
          if (value < 1.0f) value = -1.0f;
          /* all timestamps equal (one event, one-stamp bursts): t_norm = 0 / 0 = NaN; torch's t_norm.int() is INT_MIN on the
           * CPU, so the reference's mask (representations.py:94-101) drops the event -- stated here instead of relying on the
           * undefined (int)NaN */
          if (tn != tn) continue;
          const int xl = (int)x[i] + dx, yl = (int)y[i] + dy, tl = (int)tn + dt;
          if (xl < W && xl >= 0 && yl < H && yl >= 0 && tl >= 0 && tl < bins) {
            const float w = value * (1.0f - fabsf((float)xl - x[i])) * (1.0f - fabsf((float)yl - y[i])) * (1.0f - fabsf((float)tl - tn));
            grid[((size_t)tl * H + yl) * W + xl] += w;
          }
        }
  if (normalize) {
    double c = 0, s = 0, q = 0;
    for (size_t i = 0; i < per; ++i)
      if (grid[i] != 0.0f) {
        c += 1;
        s += grid[i];
        q += (double)grid[i] * grid[i];
      }
    if (c > 0) {
      const double mean = s / c;
      double var = c > 1 ? (q - c * mean * mean) / (c - 1) : 0.0;
      if (var < 0) var = 0;
      const float meanf = (float)mean, stdf = (float)sqrt(var);
      for (size_t i = 0; i < per; ++i)
        if (grid[i] != 0.0f) grid[i] = stdf > 0.0f ? (grid[i] - meanf) / stdf : (grid[i] - meanf);
    }
  }
}

/* draw_events_accumulation_image (datasets/visualize.py:23-50) followed by `> 0`
 * (test_events-image_same-time.py:137): count image, min-max scaled to 0..255 in float64,
 * truncated to uint8. */
EXPORT void orc_events_mask(const float* x, const float* y, long long n, int H, int W, uint8_t* mask) {
  double* img = (double*)calloc((size_t)H * W, sizeof(double));
  for (long long i = 0; i < n; ++i) {
    const int yi = (int)y[i], xi = (int)x[i];
    if (yi >= 0 && yi < H && xi >= 0 && xi < W) img[(size_t)yi * W + xi] += 1.0;
  }
  double lo = img[0], hi = img[0];
  for (size_t i = 0; i < (size_t)H * W; ++i) {
    if (img[i] < lo) lo = img[i];
    if (img[i] > hi) hi = img[i];
  }
  for (size_t i = 0; i < (size_t)H * W; ++i) {
    double v = (img[i] - lo) / (hi - lo) * 255.0;
    if (v > 255.0) v = 255.0;
    mask[i] = (v == v && (int)v > 0) ? 1 : 0;
  }
  free(img);
}

/* =======================================================================================
 * Evaluation metrics of the reference's harness (the step after the path).
 * ===================================================================================== */
static void orc_warp(const float* h, float x, float y, float* ox, float* oy) {
  const float a = fmaf(h[2], 1.0f, fmaf(h[1], y, h[0] * x));
  const float b = fmaf(h[5], 1.0f, fmaf(h[4], y, h[3] * x));
  const float c = fmaf(h[8], 1.0f, fmaf(h[7], y, h[6] * x));
  *ox = a / c;
  *oy = b / c;
}

/* MatchingRatio (matching_metrics.py:30-51), MeanMatchingAccuracy (:84-156),
 * ValidDescriptorsDistance (keypoints_metrics.py:160-290) for one pair.
 * k0 [n,3], k1 [m,3] keypoints ((y,x,p) if kp_yx), d0/d1 descriptors, mk0/mk1 [M,cols] matches,
 * hom 3x3 row-major or NULL (identity).  out = MR, MMA@t.., (Rep, ValidDist, Angle)@t.. */
EXPORT void orc_pair_metrics(const float* k0, int n, const float* k1, int m, const float* d0, const float* d1, int D, const float* mk0,
                             const float* mk1, int M, int cols, const float* hom, int H0, int W0, int H1, int W1, int kp_yx,
                             const float* mma_thr, int n_mma, const float* vdd_thr, int n_vdd, double* out) {
  float h[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, hi[9];
  if (hom) memcpy(h, hom, sizeof(h));
  {
    const float c00 = h[4] * h[8] - h[5] * h[7], c01 = h[5] * h[6] - h[3] * h[8], c02 = h[3] * h[7] - h[4] * h[6];
    const float det = h[0] * c00 + h[1] * c01 + h[2] * c02;
    hi[0] = c00 / det; hi[1] = (h[2] * h[7] - h[1] * h[8]) / det; hi[2] = (h[1] * h[5] - h[2] * h[4]) / det;
    hi[3] = c01 / det; hi[4] = (h[0] * h[8] - h[2] * h[6]) / det; hi[5] = (h[2] * h[3] - h[0] * h[5]) / det;
    hi[6] = c02 / det; hi[7] = (h[1] * h[6] - h[0] * h[7]) / det; hi[8] = (h[0] * h[4] - h[1] * h[3]) / det;
  }
  const int xi = kp_yx ? 1 : 0, yi = kp_yx ? 0 : 1;
  int o = 0;
  out[o++] = (double)M / ((double)(n < m ? n : m) + 1e-8);
  for (int t = 0; t < n_mma; ++t) {
    int good = 0;
    for (int i = 0; i < M; ++i) {
      float wx, wy;
      orc_warp(h, mk0[i * cols + xi], mk0[i * cols + yi], &wx, &wy);
      const float dx = wx - mk1[i * cols + xi], dy = wy - mk1[i * cols + yi];
      if (sqrtf(dx * dx + dy * dy) <= mma_thr[t]) ++good;
    }
    out[o++] = M > 0 ? (double)((float)good / (float)M) : 0.0;
  }
  float* tw = (float*)malloc(sizeof(float) * 2 * (n + 1));
  uint8_t* keep0 = (uint8_t*)malloc(n + 1);
  uint8_t* keep1 = (uint8_t*)malloc(m + 1);
  int N1 = 0, N2 = 0;
  for (int i = 0; i < n; ++i) {
    orc_warp(h, k0[i * 3 + xi], k0[i * 3 + yi], &tw[2 * i], &tw[2 * i + 1]);
    keep0[i] = tw[2 * i] >= 0.0f && tw[2 * i] < (float)W1 && tw[2 * i + 1] >= 0.0f && tw[2 * i + 1] < (float)H1;
    N1 += keep0[i];
  }
  for (int j = 0; j < m; ++j) {
    float wx, wy;
    orc_warp(hi, k1[j * 3 + xi], k1[j * 3 + yi], &wx, &wy);
    keep1[j] = wx >= 0.0f && wx < (float)W0 && wy >= 0.0f && wy < (float)H0;
    N2 += keep1[j];
  }
  for (int t = 0; t < n_vdd; ++t) {
    int cnt = 0;
    double sd = 0.0, sa = 0.0;
    for (int side = 0; side < 2; ++side) {
      const int ns = side == 0 ? n : m, no = side == 0 ? m : n;
      for (int i = 0; i < ns; ++i) {
        if (!(side == 0 ? keep0[i] : keep1[i])) continue;
        float best = INFINITY;
        int bj = -1;
        for (int j = 0; j < no; ++j) {
          if (!(side == 0 ? keep1[j] : keep0[j])) continue;
          const int a0 = side == 0 ? i : j, b1 = side == 0 ? j : i;
          const float dx = tw[2 * a0] - k1[b1 * 3 + xi], dy = tw[2 * a0 + 1] - k1[b1 * 3 + yi];
          const float dd = sqrtf(fmaf(dy, dy, dx * dx));
          if (dd < best) {
            best = dd;
            bj = j;
          }
        }
        if (bj < 0 || !(best <= vdd_thr[t])) continue;
        const float* v1 = d0 + (size_t)(side == 0 ? i : bj) * D;
        const float* v2 = d1 + (size_t)(side == 0 ? bj : i) * D;
        float sq = 0, dot = 0, n1 = 0, n2 = 0;
        for (int c = 0; c < D; ++c) {
          const float df = v1[c] - v2[c];
          sq = fmaf(df, df, sq);
          dot = fmaf(v1[c], v2[c], dot);
          n1 = fmaf(v1[c], v1[c], n1);
          n2 = fmaf(v2[c], v2[c], n2);
        }
        ++cnt;
        sd += sqrtf(sq);
        sa += einx_acosf(dot / (sqrtf(n1) * sqrtf(n2))) * 57.29577951308232f;
      }
    }
    double rep = 0, vd = 0, ang = 0;
    if (N1 != 0 && N2 != 0) {
      rep = (double)((float)cnt / (float)(N1 + N2));
      vd = sd / (double)cnt;
      ang = sa / (double)cnt;
    }
    out[o++] = rep;
    out[o++] = vd;
    out[o++] = ang;
  }
  free(tw);
  free(keep0);
  free(keep1);
}


/* ------------------------------------------------------------------------------------------
 * The numeric contract itself (include/einx_math.h), exposed so that the CPU suite can cross-check every function against
 * libm in float64 (tests/test_oracle_golden.py::test_math_contract_vs_libm): kernels and oracle compile the SAME header, so an
 * error in it is common-mode and invisible to every GPU-vs-oracle array_equal.  fn: 0 exp, 1 log, 2 sin, 3 cos, 4 erf,
 * 5 sigmoid, 6 logsigmoid, 7 gelu, 8 acos.
 * ---------------------------------------------------------------------------------------- */
EXPORT int orc_math_eval(int fn, const float* x, long long n, float* y) {
  for (long long i = 0; i < n; ++i) {
    float sn, cs;
    switch (fn) {
      case 0: y[i] = einx_expf(x[i]); break;
      case 1: y[i] = einx_logf(x[i]); break;
      case 2: einx_sincosf(x[i], &sn, &cs); y[i] = sn; break;
      case 3: einx_sincosf(x[i], &sn, &cs); y[i] = cs; break;
      case 4: y[i] = einx_erff(x[i]); break;
      case 5: y[i] = einx_sigmoidf(x[i]); break;
      case 6: y[i] = einx_logsigmoidf(x[i]); break;
      case 7: y[i] = einx_geluf(x[i]); break;
      case 8: y[i] = einx_acosf(x[i]); break;
      default: return -1;
    }
  }
  return 0;
}
