/* A plain-C consumer of include/rrtmgp_hip.h: what a `ccall` / cgo / FFI binding sees.  Compiled as C99 by
 * tests/test_abi.py and run WITHOUT a GPU: it checks the struct sizes a binding would mirror, the host-callable McICA
 * stream, and that a solve without a device fails with a status and a message instead of crashing.
 *   gcc -std=c99 -Iinclude examples/c_consumer.c -Lrrtmgp.jl_amd -lhip_rrtmgp -Wl,-rpath,$PWD/rrtmgp.jl_amd -o c_consumer */
#include <stdio.h>
#include <string.h>

#include "rrtmgp_hip.h"

int main(void) {
    char msg[256];
    int bad = 0;
    /* struct mirrors, in the library's numbering (rrtmgp_hip_abi_sizeof) */
    const size_t sizes[] = {sizeof(rrtmgp_minor_desc),        sizeof(rrtmgp_gas_lookup_desc), sizeof(rrtmgp_cloud_lookup_desc),
                            sizeof(rrtmgp_aerosol_lookup_desc), sizeof(rrtmgp_atmos_state),    sizeof(rrtmgp_lw_bcs),
                            sizeof(rrtmgp_sw_bcs),            sizeof(rrtmgp_flux_out),        sizeof(rrtmgp_solve_opts),
                            sizeof(rrtmgp_gray_state),        sizeof(rrtmgp_params),          sizeof(rrtmgp_prepare_opts)};
    for (int i = 0; i < (int)(sizeof sizes / sizeof sizes[0]); i++)
        if (rrtmgp_hip_abi_sizeof(i) != (int)sizes[i]) {
            printf("struct %d: header says %zu bytes, library %d\n", i, sizes[i], rrtmgp_hip_abi_sizeof(i));
            bad++;
        }
    /* the McICA stream is a pure function of (seed, column, g-point, band set, draw) */
    const double u = rrtmgp_hip_mcica_uniform(42u, 7, 3, 0, 1);
    if (!(u >= 0.0 && u < 1.0) || u != rrtmgp_hip_mcica_uniform(42u, 7, 3, 0, 1) || u == rrtmgp_hip_mcica_uniform(42u, 7, 3, 1, 1)) bad++;
    /* no device: a negative status and a message */
    if (rrtmgp_hip_device_count() <= 0) {
        rrtmgp_workspace *ws = NULL;
        const int rc = rrtmgp_hip_workspace_create(0, 16, 8, RRTMGP_F32, &ws);
        if (rc >= 0 || ws != NULL) bad++;
        if (rrtmgp_hip_last_error(msg, sizeof msg) != 0 || strlen(msg) == 0) bad++;
    }
    printf("%s %s: %d problem(s)\n", "libhip_rrtmgp", rrtmgp_hip_version(), bad);
    return bad;
}
