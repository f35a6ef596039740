/* Plain-C consumer of include/ecb.h: proves the boundary is a C ABI (no C++/torch types) and that argument checking
 * works without a GPU.  Built and run by tests/test_host_logic.py::test_plain_c_program_links_against_the_abi. */
#include <stdio.h>
#include <string.h>
#include "ecb.h"

int main(void) {
    if (ecb_abi_version() != ECB_ABI_VERSION) return 1;
    ecb_config cfg;
    memset(&cfg, 0, sizeof cfg);
    ecb_handle* h = NULL;
    cfg.struct_size = 4;                                  /* wrong on purpose */
    if (ecb_create(&cfg, &h) != ECB_ERR_ARG || h != NULL) return 2;
    cfg.struct_size = sizeof cfg; cfg.n_loci = 10; cfg.n_haplotypes = 40;   /* > 31 haplotypes */
    if (ecb_create(&cfg, &h) != ECB_ERR_ARG) return 3;
    cfg.n_haplotypes = 2;
    int rc = ecb_create(&cfg, &h);
    if (ecb_device_count() == 0) {                        /* no GPU: must fail loudly, never fall back */
        if (rc != ECB_ERR_NO_DEVICE || h != NULL) return 4;
        if (!strstr(ecb_last_error(NULL), "no CPU path")) return 5;
        printf("abi ok (no device)\n");
        return 0;
    }
    if (rc != ECB_OK) return 6;
    uint32_t rid[3] = {0, 0, 1}, loc[3] = {1, 1, 2}, hf[3] = {0, 1u << 16, 0};
    ecb_sizes s;
    if (ecb_push(h, rid, loc, hf, NULL, 3) != ECB_OK || ecb_finalize(h, &s) != ECB_OK) return 7;
    if (s.n_ecs != 2 || s.valid_alignments != 3 || s.n_reads != 2) return 8;
    ecb_destroy(h);
    printf("abi ok (device)\n");
    return 0;
}
