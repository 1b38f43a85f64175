/* abi_check.c -- include/cavmd.h consumed as plain C99 (what a cgo / JNI / ctypes-generator style binding sees).
 * Built and run by tests/test_capi_abi.py::test_header_is_plain_c_and_links; needs no GPU: it only touches the entry points
 * that work without a device (parameters, the thermostat's scalar rule, version, error strings) and checks that a compute
 * call without a device is refused with an error code, never served by a fallback. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "cavmd.h"

int main(void)
{
    cavmd_params p = cavmd_make_params(2000.0 / 219474.63, 1e-3, 1.0);
    if (p.K != 1.0 * (2000.0 / 219474.63) * (2000.0 / 219474.63))
        return 1;
    if (sizeof(cavmd_double4) != 32 || sizeof(cavmd_int3) != 12 || sizeof(cavmd_params) != 32 || sizeof(cavmd_result) != 192
        || sizeof(cavmd_bussi_reservoir) != 32)
        return 2;
    if (cavmd_version() != CAVMD_VERSION_MAJOR * 1000 + CAVMD_VERSION_MINOR)
        return 3;
    if (strcmp(cavmd_error_string(CAVMD_OK), "success") != 0 || strlen(cavmd_error_string(CAVMD_ERR_SYNC_TIMEOUT)) == 0)
        return 4;
    {
        double a = 0.0;
        /* tau = 0, Nf = 1: alpha = sign(R) |R| sqrt(kT / 2K) */
        if (cavmd_bussi_rescale_factor(3.0, 1.0, 0.01, 1.5, 0.0, -0.5, 0.0, &a) != CAVMD_OK || !(a < 0.0)
            || fabs(a * a * 3.0 - 1.5 * 0.25 / 2.0) > 1e-15)
            return 5;
    }
    {
        cavmd_bussi_reservoir st = {0.0, 0.0, 0.0, 0.0};
        const double var[4] = {0.1, 140.0, 0.0, 0.0};
        double f[2];
        if (cavmd_bussi_step(&st, 2.0, 297.0, 0.0, 0.0, 0.01, 1.5, 0.2, var, f) != CAVMD_OK || f[1] != 1.0
            || st.reservoir_translational != 2.0 * (1.0 - f[0] * f[0]))
            return 6;
        if (cavmd_bussi_step(&st, 0.0, 3.0, 0.0, 0.0, 0.01, 1.5, 0.2, var, f) != CAVMD_ERR_BAD_PARAMS)
            return 7;
    }
    {
        /* null arguments are refused before anything else (src/CavityForceComputeGPU.cu:522-528) */
        if (cavmd_compute_hoomd(NULL, NULL, 10, NULL, NULL, NULL, 1.0, 1.0, 1.0, 2, &p, NULL) != CAVMD_ERR_INVALID_VALUE)
            return 8;
        if (cavmd_create(-1, 1000, NULL) != CAVMD_ERR_INVALID_VALUE)
            return 9;
    }
    {
        cavmd_workspace* ws = NULL;
        const int st = cavmd_create(-1, 1000, &ws);
        if (st == CAVMD_OK)
        {
            printf("device present\n");
            cavmd_destroy(ws);
        }
        else if (st == CAVMD_ERR_NO_DEVICE && ws == NULL)
            printf("no device: refused with CAVMD_ERR_NO_DEVICE\n");
        else
            return 10;
    }
    printf("C-ABI-OK\n");
    return 0;
}
