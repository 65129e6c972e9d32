// Host-side loaders (RGL tensor file, .bsdfw weight file) driven under the address sanitizer on the CPU build (tools/asan/run_loaders_asan.py): every argument is
// handed to both loaders; malformed files must come back as BSDFD_EIO without an ASan report.
// GPU ASan is not available on this pool, so only the parsers — which run before any device call — are covered.
#include <cstdio>
#include <cstdlib>
#include "bsdfd.h"
int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        bsdfd_measured_handle h = nullptr;
        int rc = bsdfd_measured_create_from_file(argv[i], &h);
        std::printf("%s -> rc=%d %s\n", argv[i], rc, rc ? bsdfd_last_error() : "ok");
        if (h) bsdfd_measured_destroy(h);
        bsdfd_handle f = nullptr;
        rc = bsdfd_create_from_file(argv[i], 0, &f);
        std::printf("  as weights -> rc=%d\n", rc);
        if (f) bsdfd_destroy(f);
    }
    return 0;
}
