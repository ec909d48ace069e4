// logmel_host.cpp -- runs the per-lane phases of audio_tokens_amd/csrc/logmel_core.h on the host,
// 16 "lanes" one after the other, so tests can check the FFT index algebra without a GPU.
// TEST INFRASTRUCTURE (built on the fly by tests/test_logmel_core_host.py with g++).
#include <cmath>
#include <vector>

#include "../../audio_tokens_amd/csrc/logmel_core.h"

using namespace logmel;

extern "C" void logmel_host_power(const float* frame /*512*/, const float* win /*512*/, float* power /*257*/) {
    std::vector<float> tw256(512), tw512(512), ebuf(FRAME_LDS_FLOATS), zbuf(512);
    for (int j = 0; j < 256; j++) {
        const int e = (j / 16) * (j % 16);  // [k1][m2] -> W256^(m2*k1)
        tw256[2 * j] = (float)std::cos(2.0 * M_PI * e / 256.0);
        tw256[2 * j + 1] = (float)-std::sin(2.0 * M_PI * e / 256.0);
        tw512[2 * j] = (float)std::cos(2.0 * M_PI * j / 512.0);
        tw512[2 * j + 1] = (float)-std::sin(2.0 * M_PI * j / 512.0);
    }
    cpx z[16][16];
    for (int l = 0; l < 16; l++) phase1(l, frame, win, tw256.data(), ebuf.data());
    for (int l = 0; l < 16; l++) phase2(l, ebuf.data(), z[l]);
    for (int l = 0; l < 16; l++) phase3_publish(l, z[l], zbuf.data());
    for (int l = 0; l < 16; l++) {
        float p[16], p256 = 0.f;
        phase3_power(l, z[l], zbuf.data(), tw512.data(), p, p256);
        for (int e = 0; e < 16; e++) power[l + 16 * e] = p[e];
        if (l == 0) power[256] = p256;
    }
}

extern "C" void logmel_host_dft16(float* re_im /*32*/) {
    cpx v[16];
    for (int i = 0; i < 16; i++) v[i] = {re_im[2 * i], re_im[2 * i + 1]};
    dft16(v);
    for (int i = 0; i < 16; i++) { re_im[2 * i] = v[i].re; re_im[2 * i + 1] = v[i].im; }
}
