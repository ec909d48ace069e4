// logmel_core.h -- per-frame arithmetic of the fused STFT -> power -> mel -> dB kernel.
//
// Everything here is written from the point of view of ONE of the 16 lanes that share a frame
// (lane id `l` in 0..15) and only touches memory through the pointers it is given, so the very
// same code runs inside the HIP kernel (pointers into LDS) and, lane after lane, in a host test
// harness (tests/host_harness) that checks the index algebra against numpy without a GPU.
//
// A 512-point real FFT is done as a 256-point complex FFT of z[m] = x[2m] + i*x[2m+1], itself
// split 16 x 16: phase 1 = 16-point FFTs down the columns (m = 16*m1 + m2, lane = m2) with the
// W256^(m2*k1) twiddle, one transpose through LDS, phase 2 = 16-point FFTs along the rows
// (lane = k1), then the usual even/odd untangling to X[0..256] and |X|^2.
#pragma once

#if defined(__HIPCC__)
#define AT_HD __host__ __device__ __forceinline__
#else
#define AT_HD inline
#endif

namespace logmel {

constexpr int NFFT = 512;
constexpr int NBIN = NFFT / 2 + 1;     // 257
constexpr int M = NFFT / 2;            // 256-point complex FFT
constexpr int EPITCH = 18;             // complex elements per row of the transpose buffer (144 B:
                                       // 16-byte aligned and conflict-free for 128-bit row reads)
constexpr int FRAME_LDS_FLOATS = 16 * EPITCH * 2;  // 576 floats = 2304 B per frame (9 bank rows)

struct cpx {
    float re, im;
};

AT_HD cpx cadd(cpx a, cpx b) { return {a.re + b.re, a.im + b.im}; }
AT_HD cpx csub(cpx a, cpx b) { return {a.re - b.re, a.im - b.im}; }
// Fused multiply-adds are written out (the library is compiled with -ffp-contract=off so that
// the bit-exact kernels elsewhere are not at the optimiser's mercy).
AT_HD cpx cmul(cpx a, cpx w) {
    return {__builtin_fmaf(a.re, w.re, -(a.im * w.im)), __builtin_fmaf(a.re, w.im, a.im * w.re)};
}

// forward DFT of 4 points, outputs in natural order
AT_HD void dft4(cpx& a, cpx& b, cpx& c, cpx& d) {
    const cpx t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    c = csub(t0, t2);
    b = {t1.re + t3.im, t1.im - t3.re};  // t1 - i*t3
    d = {t1.re - t3.im, t1.im + t3.re};  // t1 + i*t3
}

// forward DFT of 16 points in registers, natural order in and out (4 x 4 Cooley-Tukey).
AT_HD void dft16(cpx (&v)[16]) {
    // W16^m = exp(-2*pi*i*m/16)
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    // inner DFT4 over n2 for each n1: x[n1 + 4*n2] -> A[n1][k2] stored back at v[n1 + 4*k2]
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) dft4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
    // twiddle A[n1][k2] *= W16^(n1*k2)
    v[1 + 4 * 1] = cmul(v[1 + 4 * 1], {C1, -S1});   // m=1
    v[1 + 4 * 2] = cmul(v[1 + 4 * 2], {R2, -R2});   // m=2
    v[1 + 4 * 3] = cmul(v[1 + 4 * 3], {S1, -C1});   // m=3
    v[2 + 4 * 1] = cmul(v[2 + 4 * 1], {R2, -R2});   // m=2
    v[2 + 4 * 2] = {v[2 + 4 * 2].im, -v[2 + 4 * 2].re};  // m=4: -i
    v[2 + 4 * 3] = cmul(v[2 + 4 * 3], {-R2, -R2});  // m=6
    v[3 + 4 * 1] = cmul(v[3 + 4 * 1], {S1, -C1});   // m=3
    v[3 + 4 * 2] = cmul(v[3 + 4 * 2], {-R2, -R2});  // m=6
    v[3 + 4 * 3] = cmul(v[3 + 4 * 3], {-C1, S1});   // m=9
    // outer DFT4 over n1 for each k2: A[0..3][k2] -> X[4*k1 + k2], left at v[k1 + 4*k2]
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) dft4(v[0 + 4 * k2], v[1 + 4 * k2], v[2 + 4 * k2], v[3 + 4 * k2]);
    // un-permute: v[k1 + 4*k2] holds X[4*k1 + k2]  (a 4x4 transpose)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = a + 1; b < 4; b++) {
            const cpx t = v[a + 4 * b];
            v[a + 4 * b] = v[b + 4 * a];
            v[b + 4 * a] = t;
        }
}

// Phase 1, lane l = m2: windowed samples -> column FFT -> W256 twiddle -> transpose buffer.
//   frame : the 512 samples of this frame (already reflect-padded), win : 512 window values,
//   tw256 : 16 x 16 x (cos, -sin): entry [k1][m2] = W256^(m2*k1), so the 16 lanes of a frame read
//           consecutive words; ebuf : FRAME_LDS_FLOATS floats.
// (register form: winr[2*m1 + {0,1}] = win[32*m1 + 2*l + {0,1}], twr[k1] = tw256 entry [k1][l]; a lane's
// table entries do not depend on the frame, so a kernel that walks many frames loads them once)
AT_HD void phase1_r(int l, const float* frame, const float (&winr)[32], const cpx (&twr)[16], float* ebuf) {
    cpx v[16];
#pragma unroll
    for (int m1 = 0; m1 < 16; m1++) {
        const int s = 32 * m1 + 2 * l;
        v[m1] = {frame[s] * winr[2 * m1], frame[s + 1] * winr[2 * m1 + 1]};  // torch: frames * window, fp32
    }
    dft16(v);  // v[k1]
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        const cpx y = k1 == 0 ? v[0] : cmul(v[k1], twr[k1]);
        ebuf[2 * (k1 * EPITCH + l)] = y.re;
        ebuf[2 * (k1 * EPITCH + l) + 1] = y.im;
    }
}
// (window from memory, twiddles in registers)
AT_HD void phase1_w(int l, const float* frame, const float* win, const cpx (&twr)[16], float* ebuf) {
    cpx v[16];
#pragma unroll
    for (int m1 = 0; m1 < 16; m1++) {
        const int s = 32 * m1 + 2 * l;
        v[m1] = {frame[s] * win[s], frame[s + 1] * win[s + 1]};  // torch: frames * window, fp32
    }
    dft16(v);  // v[k1]
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) {
        const cpx y = k1 == 0 ? v[0] : cmul(v[k1], twr[k1]);
        ebuf[2 * (k1 * EPITCH + l)] = y.re;
        ebuf[2 * (k1 * EPITCH + l) + 1] = y.im;
    }
}
AT_HD void load_phase1_tables(int l, const float* win, const float* tw256, float (&winr)[32], cpx (&twr)[16]) {
#pragma unroll
    for (int m1 = 0; m1 < 16; m1++) {
        winr[2 * m1] = win[32 * m1 + 2 * l];
        winr[2 * m1 + 1] = win[32 * m1 + 2 * l + 1];
    }
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) twr[k1] = {tw256[2 * (k1 * 16 + l)], tw256[2 * (k1 * 16 + l) + 1]};
}
AT_HD void phase1(int l, const float* frame, const float* win, const float* tw256, float* ebuf) {
    float winr[32];
    cpx twr[16];
    load_phase1_tables(l, win, tw256, winr, twr);
    phase1_r(l, frame, winr, twr, ebuf);
}

// Phase 2, lane l = k1: row FFT.  On return z[k2] = Z[k1 + 16*k2].
AT_HD void phase2(int l, const float* ebuf, cpx (&z)[16]) {
#pragma unroll
    for (int m2 = 0; m2 < 16; m2++) z[m2] = {ebuf[2 * (l * EPITCH + m2)], ebuf[2 * (l * EPITCH + m2) + 1]};
    dft16(z);
}

// Phase 3a, lane l = k1: publish Z in natural order, zbuf[2*k], k = 0..255 (512 floats; may alias
// the transpose buffer once every lane of the frame has finished phase 2).
AT_HD void phase3_publish(int l, const cpx (&z)[16], float* zbuf) {
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
        zbuf[2 * (l + 16 * k2)] = z[k2].re;
        zbuf[2 * (l + 16 * k2) + 1] = z[k2].im;
    }
}

// Phase 3b, lane l: power of bins k = l + 16*e (e = 0..15); lane 0 also returns bin 256 in p256.
//   tw512 : 256 x (cos, -sin) of 2*pi*k/512.
AT_HD void phase3_power_r(int l, const cpx (&z)[16], const float* zbuf, const cpx (&t5)[16], float (&p)[16],
                          float& p256) {
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int k = l + 16 * e;
        const int kk = (M - k) & (M - 1);             // Z[256-k], with Z[256] == Z[0]
        const cpx a = z[e];
        const cpx b = {zbuf[2 * kk], -zbuf[2 * kk + 1]};  // conj(Z[M-k])
        const cpx ev = {0.5f * (a.re + b.re), 0.5f * (a.im + b.im)};
        const cpx df = {0.5f * (a.re - b.re), 0.5f * (a.im - b.im)};
        const cpx od = {df.im, -df.re};               // -i * df
        const cpx x = cadd(ev, cmul(od, t5[e]));
        p[e] = __builtin_fmaf(x.re, x.re, x.im * x.im);
        if (k == 0) {
            const float n = a.re - a.im;              // X[256] = Re Z0 - Im Z0 (purely real)
            p256 = n * n;
        }
    }
}
AT_HD void load_phase3_table(int l, const float* tw512, cpx (&t5)[16]) {
#pragma unroll
    for (int e = 0; e < 16; e++) t5[e] = {tw512[2 * (l + 16 * e)], tw512[2 * (l + 16 * e) + 1]};
}
AT_HD void phase3_power(int l, const cpx (&z)[16], const float* zbuf, const float* tw512,
                        float (&p)[16], float& p256) {
    cpx t5[16];
    load_phase3_table(l, tw512, t5);
    phase3_power_r(l, z, zbuf, t5, p, p256);
}

// Phase 4: one mel filter = banded dot product over the power spectrum.  The band is stored padded
// with zero weights to whole, 4-aligned quads of bins (start4 % 4 == 0, n4 quads), so the device
// reads power and weights 16 bytes at a time; a zero weight leaves the running sum unchanged.
AT_HD float mel_band(const float* pw, int start4, int n4, const float* wts) {
    float s = 0.0f;
    for (int q = 0; q < n4; q++) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 pv = *reinterpret_cast<const f4*>(pw + start4 + 4 * q);
        const f4 wv = *reinterpret_cast<const f4*>(wts + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; e++) s = __builtin_fmaf(pv[e], wv[e], s);
#else
        for (int e = 0; e < 4; e++) s = __builtin_fmaf(pw[start4 + 4 * q + e], wts[4 * q + e], s);
#endif
    }
    return s;
}

// The same with the band stored as it is (start, length in bins): 4-byte reads.
AT_HD float mel_band_plain(const float* pw, int start, int len, const float* wts) {
    float s = 0.0f;
    for (int w = 0; w < len; w++) s = __builtin_fmaf(pw[start + w], wts[w], s);
    return s;
}

}  // namespace logmel
