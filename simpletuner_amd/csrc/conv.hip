// conv.hip — layout passes around the convolution-as-GEMM of the UNet / VAE path (gemm.hip: st355_conv_bf16).
//
// "grid buffer" (see st355_conv_grid_rows): [B*(H+2)*(W+2) + 64, C] bf16, position (b, y, x) of the zero-bordered (H+2) x (W+2) image at row
// (b*(H+2) + y)*(W+2) + x; border positions are ZERO; the 64 tail rows are zero and are never written by any kernel (the host allocates
// grid buffers zero-filled once and re-uses them).  Everything here is a single coalesced HBM pass (16-byte accesses along C).
//   reference seams: diffusers Downsample2D (conv 3x3 stride 2 pad 1), Upsample2D (nearest 2x then conv 3x3), UNet conv_in / conv_out on
//   [B,4,H,W] latents (sdxl/model.py:350-367 calls the UNet positionally), Transformer2DModel's NCHW <-> token reshapes.
#include "common.h"

// ---- NCHW latents <-> grid (channels zero-padded to Cpad) ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_grid_from_nchw(const bf16* __restrict__ x, bf16* __restrict__ g, int B, int C, int H, int W, int Cpad) {
  const int Hp = H + 2, Wp = W + 2;
  const int64_t n = (int64_t)B * Hp * Wp * Cpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const int64_t pos = i / Cpad;
    const int xx = (int)(pos % Wp), yy = (int)((pos / Wp) % Hp), b = (int)(pos / ((int64_t)Wp * Hp));
    float v = 0.f;
    if (c < C && yy >= 1 && yy <= H && xx >= 1 && xx <= W) v = bf2f(x[(((int64_t)b * C + c) * H + (yy - 1)) * W + (xx - 1)]);
    g[i] = f2bf(v);
  }
}
__global__ void __launch_bounds__(256) k_grid_to_nchw(const bf16* __restrict__ g, bf16* __restrict__ y, int B, int C, int H, int W, int Cpad) {
  const int Wp = W + 2, Hp = H + 2;
  const int64_t n = (int64_t)B * C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W), yy = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C), b = (int)(i / ((int64_t)W * H * C));
    y[i] = g[(((int64_t)b * Hp + yy + 1) * Wp + xx + 1) * Cpad + c];
  }
}
extern "C" int st355_grid_from_nchw(void* stream, const void* x, void* grid, int B, int C, int H, int W, int Cpad) {
  ST_REQUIRE(x && grid && B > 0 && C > 0 && Cpad >= C, "grid_from_nchw: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0.0, 4.0 * B * Cpad * (H + 2) * (W + 2));
  hipLaunchKernelGGL(k_grid_from_nchw, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)grid, B, C, H, W, Cpad);
  return st355_check_launch("grid_from_nchw");
}
extern "C" int st355_grid_to_nchw(void* stream, const void* grid, void* y, int B, int C, int H, int W, int Cpad) {
  ST_REQUIRE(y && grid && B > 0 && C > 0 && Cpad >= C, "grid_to_nchw: bad args");
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0.0, 4.0 * B * C * H * W);
  hipLaunchKernelGGL(k_grid_to_nchw, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const bf16*)grid, (bf16*)y, B, C, H, W, Cpad);
  return st355_check_launch("grid_to_nchw");
}

// ---- 3x3 column gather (stride 1 or 2, pad 1) onto the OUTPUT grid: col[(b,yo,xo), tap*C + c] = x[(b, s*(yo-1)+ky, s*(xo-1)+kx), c] ------
// used only where the shifted-view GEMM does not apply: the two stride-2 Downsample2D convs, conv_in (C = 4 -> 8) and conv_out's input
// gradient (C = 4 -> 8).  One thread = one 8-channel (16-byte) chunk; columns >= 9*C (the pad to a multiple of 64) are written as zero.
__global__ void __launch_bounds__(256) k_im2col3x3(const bf16* __restrict__ x, bf16* __restrict__ col, int B, int H, int W, int C, int stride, int Kpad, int off) {
  const int Ho = H / stride, Wo = W / stride, Hop = Ho + 2, Wop = Wo + 2, Hp = H + 2, Wp = W + 2;
  const int kc = Kpad / 8, c8 = C / 8;
  const int64_t n = (int64_t)B * Hop * Wop * kc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % kc);
    const int64_t pos = i / kc;
    const int xo = (int)(pos % Wop), yo = (int)((pos / Wop) % Hop), b = (int)(pos / ((int64_t)Wop * Hop));
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = f2bf(0.f);
    const int tap = ch / c8, cc = ch - tap * c8;
    if (tap < 9 && yo >= 1 && yo <= Ho && xo >= 1 && xo <= Wo) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int yi = stride * (yo - 1) + ky + off, xi = stride * (xo - 1) + kx + off;   // padded input coordinates, always inside [0,H+1]x[0,W+1]
      v = *(const bf16x8*)(x + (((int64_t)b * Hp + yi) * Wp + xi) * C + cc * 8);
    }
    *(bf16x8*)(col + pos * Kpad + ch * 8) = v;
  }
}
// adjoint: dx[(b,yi,xi), c] = sum over taps with s*(yo-1)+ky == yi, s*(xo-1)+kx == xi of dcol[(b,yo,xo), tap*C + c]   (gather form, fixed order)
__global__ void __launch_bounds__(256) k_col2im3x3(const bf16* __restrict__ dcol, bf16* __restrict__ dx, int B, int H, int W, int C, int stride, int Kpad, int off) {
  const int Ho = H / stride, Wo = W / stride, Hop = Ho + 2, Wop = Wo + 2, Hp = H + 2, Wp = W + 2;
  const int c8 = C / 8;
  const int64_t n = (int64_t)B * Hp * Wp * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t pos = i / c8;
    const int xi = (int)(pos % Wp), yi = (int)((pos / Wp) % Hp), b = (int)(pos / ((int64_t)Wp * Hp));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    if (yi >= 1 && yi <= H && xi >= 1 && xi <= W) {
      for (int ky = 0; ky < 3; ky++) {
        const int ty = yi - ky - off;
        if (ty < 0 || ty % stride) continue;
        const int yo = ty / stride + 1;
        if (yo < 1 || yo > Ho) continue;
        for (int kx = 0; kx < 3; kx++) {
          const int tx = xi - kx - off;
          if (tx < 0 || tx % stride) continue;
          const int xo = tx / stride + 1;
          if (xo < 1 || xo > Wo) continue;
          const bf16x8 v = *(const bf16x8*)(dcol + (((int64_t)b * Hop + yo) * Wop + xo) * Kpad + (ky * 3 + kx) * C + cc * 8);
#pragma unroll
          for (int j = 0; j < 8; j++) acc[j] += bf2f(v[j]);
        }
      }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = f2bf(acc[j]);
    *(bf16x8*)(dx + pos * C + cc * 8) = o;
  }
}
extern "C" int st355_im2col3x3(void* stream, const void* x, void* col, int B, int H, int W, int C, int stride, int Kpad, int pad) {
  ST_REQUIRE(x && col && C % 8 == 0 && Kpad % 8 == 0 && Kpad >= 9 * C && (stride == 1 || stride == 2) && H % stride == 0 && W % stride == 0,
             "im2col3x3: bad args (C=%d Kpad=%d stride=%d)", C, Kpad, stride);
  ST_REQUIRE(pad == 1 || (pad == 0 && stride == 2), "im2col3x3: pad 0 is the stride-2 (0,1,0,1)-padded form only");
  const int64_t n = (int64_t)B * (H / stride + 2) * (W / stride + 2) * (Kpad / 8);
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0.0, 32.0 * n);
  hipLaunchKernelGGL(k_im2col3x3, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 65536)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)col, B, H, W, C,
                     stride, Kpad, pad ? 0 : 1);
  return st355_check_launch("im2col3x3");
}
extern "C" int st355_col2im3x3(void* stream, const void* dcol, void* dx, int B, int H, int W, int C, int stride, int Kpad, int pad) {
  ST_REQUIRE(dcol && dx && C % 8 == 0 && Kpad % 8 == 0 && Kpad >= 9 * C && (stride == 1 || stride == 2) && H % stride == 0 && W % stride == 0,
             "col2im3x3: bad args");
  const int64_t n = (int64_t)B * (H + 2) * (W + 2) * (C / 8);
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0.0, 16.0 * n * (1 + 9.0 / (stride * stride)));
  hipLaunchKernelGGL(k_col2im3x3, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 65536)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dcol, (bf16*)dx, B, H, W, C,
                     stride, Kpad, pad ? 0 : 1);
  return st355_check_launch("col2im3x3");
}

// ---- nearest 2x upsample (Upsample2D before its conv) and its adjoint -----------------------------------------------------------------
__global__ void __launch_bounds__(256) k_upsample2x(const bf16* __restrict__ x, bf16* __restrict__ y, int B, int H, int W, int C) {
  const int H2 = 2 * H, W2 = 2 * W, Hp2 = H2 + 2, Wp2 = W2 + 2, Hp = H + 2, Wp = W + 2, c8 = C / 8;
  const int64_t n = (int64_t)B * Hp2 * Wp2 * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t pos = i / c8;
    const int xx = (int)(pos % Wp2), yy = (int)((pos / Wp2) % Hp2), b = (int)(pos / ((int64_t)Wp2 * Hp2));
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = f2bf(0.f);
    if (yy >= 1 && yy <= H2 && xx >= 1 && xx <= W2) v = *(const bf16x8*)(x + (((int64_t)b * Hp + (yy - 1) / 2 + 1) * Wp + (xx - 1) / 2 + 1) * C + cc * 8);
    *(bf16x8*)(y + pos * C + cc * 8) = v;
  }
}
__global__ void __launch_bounds__(256) k_upsample2x_bwd(const bf16* __restrict__ dy, bf16* __restrict__ dx, int B, int H, int W, int C) {
  const int Wp2 = 2 * W + 2, Hp2 = 2 * H + 2, Hp = H + 2, Wp = W + 2, c8 = C / 8;
  const int64_t n = (int64_t)B * Hp * Wp * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t pos = i / c8;
    const int xx = (int)(pos % Wp), yy = (int)((pos / Wp) % Hp), b = (int)(pos / ((int64_t)Wp * Hp));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
#pragma unroll
      for (int dyy = 0; dyy < 2; dyy++)
#pragma unroll
        for (int dxx = 0; dxx < 2; dxx++) {
          const bf16x8 v = *(const bf16x8*)(dy + (((int64_t)b * Hp2 + 2 * (yy - 1) + dyy + 1) * Wp2 + 2 * (xx - 1) + dxx + 1) * C + cc * 8);
#pragma unroll
          for (int j = 0; j < 8; j++) acc[j] += bf2f(v[j]);
        }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = f2bf(acc[j]);
    *(bf16x8*)(dx + pos * C + cc * 8) = o;
  }
}
extern "C" int st355_upsample2x(void* stream, const void* x, void* y, int B, int H, int W, int C) {
  ST_REQUIRE(x && y && C % 8 == 0 && B > 0, "upsample2x: bad args");
  const int64_t n = (int64_t)B * (2 * H + 2) * (2 * W + 2) * (C / 8);
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0.0, 20.0 * n);
  hipLaunchKernelGGL(k_upsample2x, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 65536)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, B, H, W, C);
  return st355_check_launch("upsample2x");
}
extern "C" int st355_upsample2x_bwd(void* stream, const void* dy, void* dx, int B, int H, int W, int C) {
  ST_REQUIRE(dy && dx && C % 8 == 0 && B > 0, "upsample2x_bwd: bad args");
  const int64_t n = (int64_t)B * (H + 2) * (W + 2) * (C / 8);
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0.0, 80.0 * n);
  hipLaunchKernelGGL(k_upsample2x_bwd, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 65536)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (bf16*)dx, B, H, W, C);
  return st355_check_launch("upsample2x_bwd");
}

// ---- grid <-> dense tokens [B*H*W, C] (Transformer2DModel: NCHW -> (B, HW, C) and back, + the block's residual) -----------------------
// to_grid: out[(b,y,x)] = tokens[b*H*W + (y-1)*W + (x-1)] (+ residual[(b,y,x)]) on interior positions, zero on the border.
template <bool TO_GRID>
__global__ void __launch_bounds__(256) k_grid_tokens(const bf16* __restrict__ src, const bf16* __restrict__ residual, bf16* __restrict__ dst, int B, int H,
                                                    int W, int C) {
  const int Hp = H + 2, Wp = W + 2, c8 = C / 8;
  const int64_t n = TO_GRID ? (int64_t)B * Hp * Wp * c8 : (int64_t)B * H * W * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t pos = i / c8;
    if (TO_GRID) {
      const int xx = (int)(pos % Wp), yy = (int)((pos / Wp) % Hp), b = (int)(pos / ((int64_t)Wp * Hp));
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = f2bf(0.f);
      if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
        o = *(const bf16x8*)(src + (((int64_t)b * H + yy - 1) * W + xx - 1) * C + cc * 8);
        if (residual) {
          const bf16x8 r = *(const bf16x8*)(residual + pos * C + cc * 8);
#pragma unroll
          for (int j = 0; j < 8; j++) o[j] = f2bf(bf2f(o[j]) + bf2f(r[j]));
        }
      }
      *(bf16x8*)(dst + pos * C + cc * 8) = o;
    } else {
      const int xx = (int)(pos % W), yy = (int)((pos / W) % H), b = (int)(pos / ((int64_t)W * H));
      *(bf16x8*)(dst + pos * C + cc * 8) = *(const bf16x8*)(src + (((int64_t)b * Hp + yy + 1) * Wp + xx + 1) * C + cc * 8);
    }
  }
}
extern "C" int st355_tokens_to_grid(void* stream, const void* tokens, const void* residual, void* grid, int B, int H, int W, int C) {
  ST_REQUIRE(tokens && grid && C % 8 == 0 && B > 0, "tokens_to_grid: bad args");
  const int64_t n = (int64_t)B * (H + 2) * (W + 2) * (C / 8);
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0.0, (residual ? 48.0 : 32.0) * n);
  hipLaunchKernelGGL(k_grid_tokens<true>, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 65536)), dim3(256), 0, (hipStream_t)stream, (const bf16*)tokens,
                     (const bf16*)residual, (bf16*)grid, B, H, W, C);
  return st355_check_launch("tokens_to_grid");
}
extern "C" int st355_grid_to_tokens(void* stream, const void* grid, void* tokens, int B, int H, int W, int C) {
  ST_REQUIRE(tokens && grid && C % 8 == 0 && B > 0, "grid_to_tokens: bad args");
  const int64_t n = (int64_t)B * H * W * (C / 8);
  ProfScope ps(stream, ST355_K_ELEMENTWISE, 0.0, 32.0 * n);
  hipLaunchKernelGGL(k_grid_tokens<false>, dim3((unsigned)std::min<int64_t>(cdiv64(n, 256), 65536)), dim3(256), 0, (hipStream_t)stream, (const bf16*)grid,
                     (const bf16*)nullptr, (bf16*)tokens, B, H, W, C);
  return st355_check_launch("grid_to_tokens");
}
