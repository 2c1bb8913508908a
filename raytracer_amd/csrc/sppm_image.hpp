// sppm_image.hpp -- hitpoint state between ranks, the per-pass update of rule 1, image assembly, RGBE pixels.
// Part of the single translation unit sppm_kernels.hip (included there, in order; not a stand-alone header).
#pragma once

// ---- hitpoint state between ranks ------------------------------------------------------------------------
// out[h][5] = (r2, flux xyz, N) of the hitpoints this rank owns, zeros elsewhere: summed over the ranks (exactly
// one non-zero term per hitpoint, so the sum is exact) it is the state of a single-GPU run.
__global__ void k_export_state(HpArrays hp, Own own, double* __restrict__ out)
{
    const long long h = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long P = hp.P;
    if (h >= P) return;
    const bool mine = hp.valid[h] != 0 && owned(own, (int)h);
    double* o = out + 5 * h;
    o[0] = mine ? hp.r2[h] : 0.0;
    o[1] = mine ? hp.flux[h] : 0.0; o[2] = mine ? hp.flux[P + h] : 0.0; o[3] = mine ? hp.flux[2 * P + h] : 0.0;
    o[4] = mine ? (double)hp.n[h] : 0.0;
}
__global__ void k_import_state(HpArrays hp, const double* __restrict__ in)
{
    const long long h = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long P = hp.P;
    if (h >= P) return;
    if (!hp.valid[h]) return;
    const double* o = in + 5 * h;
    hp.r2[h] = o[0];
    hp.flux[h] = o[1]; hp.flux[P + h] = o[2]; hp.flux[2 * P + h] = o[3];
    hp.n[h] = (int32_t)o[4];
}

// image assembly, src/sppm2.h:658-670 (rule 0) / src/stochastic_ppm.h:743-765 (rule 1)
__global__ void k_resolve(HpArrays hp, int pass, int64_t photon_num, int rule, double* __restrict__ image)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int P = hp.P;
    if (i >= P) return;
    d3 c = mk(hp.ld[i], hp.ld[P + i], hp.ld[2 * P + i]) / (double)(pass + 1);
    if (hp.valid[i]) {
        if (rule == 0)
            c = c + mk(hp.flux[i], hp.flux[P + i], hp.flux[2 * P + i]) * (1.0 / (SPPM_PI * hp.r2[i] * (pass + 1) * (double)photon_num));
        else
            c = c + mk(hp.tau[i], hp.tau[P + i], hp.tau[2 * P + i]) * (1.0 / (SPPM_PI * hp.r2[i] * (double)((int64_t)(pass + 1) * photon_num)));
    }
    image[3 * (size_t)i] = c.x; image[3 * (size_t)i + 1] = c.y; image[3 * (size_t)i + 2] = c.z;
}

// get_hdr_pixel, src/hdr.h:37-45
__device__ __host__ inline void hdr_pixel(double x, double y, double z, uint8_t* out)
{
    const double m0 = y > z ? y : z;
    double d = x > m0 ? x : m0;
    if (d <= 1e-32) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    int e;
    const double m = frexp(d, &e);
    d = m * 256.0 / d;
    out[0] = (uint8_t)(int)(x * d); out[1] = (uint8_t)(int)(y * d); out[2] = (uint8_t)(int)(z * d); out[3] = (uint8_t)(e + 128);
}
__global__ void k_hdr_pixels(const double* __restrict__ rgb, int64_t n, uint8_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t px[4];
    hdr_pixel(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], px);
    out[4 * i] = px[0]; out[4 * i + 1] = px[1]; out[4 * i + 2] = px[2]; out[4 * i + 3] = px[3];
}
