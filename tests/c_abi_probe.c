/* Plain-C client of include/ttn.h: proves the header is valid C99 and the entry points link and run the way a Julia
 * `ccall` (or cgo, or any C FFI) would use them.  Only calls that need no GPU: ttn_version, ttn_r_and_d_to_rks
 * (the reference's known answers, test/test_tt_tools.jl:945-946) and one call that must refuse with TTN_ERR_NOT_INIT. */
#include <stdio.h>
#include "ttn.h"

static int show(const int64_t* dims, int64_t rmax) {
    const int64_t rks[3] = {5, 5, 5};
    int64_t out[3] = {0, 0, 0};
    const int rc = ttn_r_and_d_to_rks(2, dims, 3, rks, rmax, out);
    if (rc != TTN_OK) { printf("ttn_r_and_d_to_rks failed: %d (%s)\n", rc, ttn_last_error_string()); return 1; }
    printf("rks = %lld %lld %lld\n", (long long)out[0], (long long)out[1], (long long)out[2]);
    return 0;
}

int main(void) {
    const int64_t d1[2] = {0, 2}, d2[2] = {0, 0};
    const int64_t dims[2] = {2, 2}, cap[3] = {1, 2, 1};
    ttn_tt_t h = 0;
    printf("%s\n", ttn_version());
    if (show(d1, 4) || show(d2, 4)) return 1;
    printf("not-init rc = %d\n", ttn_tt_create(2, dims, cap, 1, &h));     /* no ttn_init: must refuse, not crash */
    return 0;
}
