/* Compiled by tests/test_abi.py with a plain C compiler: the header must be valid C99 and every
 * entry point must link.  With no GPU the calls must FAIL cleanly (no fallback). */
#include <stdio.h>
#include <string.h>
#include "ldpc_hip.h"

int main(void) {
    int32_t row_ptr[3] = {0, 2, 4}, col_idx[4] = {0, 1, 1, 2};
    ldpc_code *c = ldpc_code_create_csr(2, 3, row_ptr, col_idx);
    if (!c) { printf("code_create failed: %s\n", ldpc_last_error()); return 1; }
    int M, N, E;
    if (ldpc_code_dims(c, &M, &N, &E) != LDPC_OK || M != 2 || N != 3 || E != 4) return 2;
    int32_t off[4] = {0, -1, 1, 3};
    ldpc_code *q = ldpc_code_create_qc(4, 2, 2, off);
    if (!q) return 3;
    ldpc_code_dims(q, &M, &N, &E);
    if (M != 8 || N != 8 || E != 12) return 4;
    if (ldpc_code_create_qc(4, 1, 1, (int32_t[]){7}) != NULL || ldpc_last_error_code() != LDPC_EINVAL) return 5;
    printf("abi %d devices %d\n", ldpc_abi_version(), ldpc_device_count());
    if (ldpc_device_count() == 0) {
        if (ldpc_init(0) != LDPC_ENODEVICE) return 6;
        if (ldpc_ctx_create(c, LDPC_MINSUM, LDPC_F32, 4) != NULL || ldpc_last_error_code() != LDPC_ENODEVICE) return 7;
        if (ldpc_ecc_create("codes", "ldpc/hip-minsum/jpl.1024.4.5/50/4/5", 4) != NULL) return 8;
        if (ldpc_ecc_create("codes", "ldpc/reference/x/5", 4) != NULL || ldpc_last_error_code() != LDPC_ENOTFOUND) return 9;
    }
    ldpc_code_destroy(q);
    ldpc_code_destroy(c);
    printf("ok\n");
    return 0;
}
