// Host-side weight-image builders of the 32-query-tile kernels (csrc/flow32.hip: build_image32_t<DISK / SPHERICAL>, build_image32w)
// under the address sanitizer on the CPU build (tools/asan/run_image_asan.py): they fill compile-time-laid-out images with
// hand-computed indices, so an out-of-range index would silently corrupt the heap.  No device call is made.
#include <cstdio>
#include <cstring>
#include <vector>
#include "bsdfd.h"
#include "flow32.h"

static std::vector<char> slurp(const char* path) {
    std::vector<char> raw;
    FILE* f = std::fopen(path, "rb");
    if (!f) return raw;
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + n);
    std::fclose(f);
    return raw;
}

int main(int argc, char** argv) {   // args: <weights.bsdfw> <precision> ...
    int bad = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::vector<char> raw = slurp(argv[i]);
        const int prec = std::atoi(argv[i + 1]);
        if (raw.size() < 104 || std::memcmp(raw.data(), "BSDFWT01", 8) != 0) { std::printf("%s: not a weight file\n", argv[i]); bad = 1; continue; }
        int32_t hdr[8];
        std::memcpy(hdr, raw.data() + 72, sizeof hdr);
        bsdfd_desc d;
        std::memset(&d, 0, sizeof d);
        d.domain = hdr[0]; d.width = hdr[1]; d.n_hidden = hdr[2]; d.pe_bands = hdr[3]; d.base_hidden = hdr[4]; d.base_pe_bands = hdr[5];
        const size_t sd = d.domain == BSDFD_DOMAIN_DISK ? 2 : 3, in_dim = sd + 1 + 2 + 4 * d.pe_bands, bin = 2 + 4 * d.base_pe_bands;
        const size_t cnt[7] = {(size_t)d.width * in_dim, (size_t)(d.n_hidden - 1) * d.width * d.width, (size_t)2 * d.width,
                               (size_t)d.base_hidden * bin, (size_t)d.base_hidden, (size_t)4 * d.base_hidden, 4};
        // exact-size heap copies: a read past the end of any weight array is an ASan report as well
        std::vector<std::vector<float>> w(7);
        const float* p = reinterpret_cast<const float*>(raw.data() + 104);
        for (int k = 0; k < 7; ++k) { w[k].assign(p, p + cnt[k]); p += cnt[k]; }
        d.w_in = w[0].data(); d.w_hidden = w[1].data(); d.w_out = w[2].data();
        d.base_w1 = w[3].data(); d.base_b1 = w[4].data(); d.base_w2 = w[5].data(); d.base_b2 = w[6].data();
        if (!bsdfd_tile32_supported(d, prec)) { std::printf("%s prec %d: no 32-query-tile kernel\n", argv[i], prec); continue; }
        const std::vector<char> img = bsdfd_build_image32(d, prec);
        unsigned long long sum = 0;
        for (unsigned char c : img) sum += c;
        int lds[3], thr[3];
        for (int m = 0; m < 3; ++m) { lds[m] = bsdfd_kernel32(d, prec, m) ? bsdfd_kernel32_lds_bytes(d, prec, m) : 0; thr[m] = bsdfd_kernel32_threads(d, prec, m); }
        std::printf("%s prec %d: image %zu B (byte sum %llu), LDS per mode %d %d %d, threads %d %d %d\n", argv[i], prec, img.size(), sum,
                    lds[0], lds[1], lds[2], thr[0], thr[1], thr[2]);
        if (img.empty() || (size_t)(lds[0] > lds[1] ? lds[0] : lds[1]) < img.size()) bad = 1;
    }
    return bad;
}
