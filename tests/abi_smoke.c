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
    /* the result as host arrays, then the file conversions on host arrays too (ec2emase / emase2ec: no device allocator on this side) */
    int32_t ia[3], ja[2], da[2], in_[2], jn[2], dn[2];
    if (ecb_export(h, ia, ja, da, in_, jn, dn) != ECB_OK) return 9;          /* EC 0 = {locus 1: haplotypes A|B}, EC 1 = {locus 2: A} */
    if (ia[0] != 0 || ia[1] != 1 || ia[2] != 2 || ja[0] != 1 || da[0] != 3 || ja[1] != 2 || da[1] != 1) return 10;
    uint64_t total = 0, nnz = 0;
    int32_t cptr[2 * 11], cidx[3], ia2[3], ja2[3], da2[3];
    if (ecb_csr_to_hapcsc(0, 2, 10, 2, ia, ja, da, NULL, NULL, 0, &total) != ECB_OK || total != 3) return 11;
    if (ecb_csr_to_hapcsc(0, 2, 10, 2, ia, ja, da, cptr, cidx, 3, &total) != ECB_OK || total != 3) return 12;
    if (cptr[1] != 0 || cptr[2] != 1 || cptr[3] != 2 || cptr[10] != 2 || cptr[11 + 2] != 1 || cptr[11 + 10] != 1) return 13;   /* haplotype A: columns 1, 2; B: column 1 */
    if (cidx[0] != 0 || cidx[1] != 1 || cidx[2] != 0) return 14;
    if (ecb_hapcsc_to_csr(0, 2, 10, 2, cptr, cidx, total, ia2, ja2, da2, &nnz) != ECB_OK || nnz != 2) return 15;
    if (memcmp(ia, ia2, sizeof ia) || memcmp(ja, ja2, sizeof ja) || memcmp(da, da2, sizeof da)) return 16;
    /* the merge of two shards inside the library (both on this device here): the same two ECs */
    ecb_handle *a = NULL, *b = NULL, *root = NULL;
    if (ecb_create(&cfg, &a) != ECB_OK || ecb_create(&cfg, &b) != ECB_OK || ecb_create(&cfg, &root) != ECB_OK) return 17;
    uint32_t r0[2] = {0, 0}, r1[1] = {0};
    if (ecb_push(a, r0, loc, hf, NULL, 2) != ECB_OK || ecb_push(b, r1, loc + 2, hf + 2, NULL, 1) != ECB_OK) return 18;
    ecb_handle* both[2] = {a, b};
    ecb_sizes m;
    if (ecb_merge(both, 2, root, &m) != ECB_OK) { fprintf(stderr, "%s\n", ecb_last_error(root)); return 19; }
    if (m.n_ecs != 2 || m.n_reads != 2 || m.valid_alignments != 3) return 20;
    if (ecb_export(root, ia2, ja2, da2, NULL, NULL, NULL) != ECB_OK || memcmp(ia, ia2, sizeof ia) || memcmp(ja, ja2, sizeof ja) || memcmp(da, da2, sizeof da)) return 21;
    ecb_destroy(a); ecb_destroy(b); ecb_destroy(root);
    ecb_destroy(h);
    printf("abi ok (device)\n");
    return 0;
}
