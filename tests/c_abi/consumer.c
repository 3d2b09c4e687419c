/* A plain-C consumer of include/mi_nerf.h: proves that the header is valid C99 (no C++ in the boundary), that libmi_nerf.so links from C with
 * nothing but the header, and that the argument checks answer before any GPU call (this program runs on a box without a GPU).
 * Built and run by tests/test_c_abi_consumer_cpu.py:  gcc -std=c99 -Wall -Werror -I include consumer.c -L pkg -lmi_nerf -Wl,-rpath,pkg */
#include <stdio.h>
#include <string.h>
#include "mi_nerf.h"

#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "FAILED line %d: %s (last error: %s)\n", __LINE__, #cond, mi_nerf_last_error()); return 1; } } while (0)

int main(void) {
    EXPECT(mi_nerf_abi_version() == MI_NERF_ABI_VERSION);
    mi_nerf_net net = {8, 256, 4, 10, 4};
    const size_t blob = mi_nerf_packed_bytes(&net);
    EXPECT(blob > 2u * 1000 * 1000 && blob < 3u * 1000 * 1000);                 /* 595 844 parameters + padding */
    EXPECT(mi_nerf_param_count(&net) == 595844);                                /* SURVEY 8(a) a9 */
    mi_nerf_net narrow = {8, 64, 4, 10, 4}, wide = {8, 512, 4, 10, 4}, too_wide = {8, 600, 4, 10, 4};
    mi_nerf_net w128 = {8, 128, 4, 10, 4};
    EXPECT(mi_nerf_packed_bytes(&narrow) == mi_nerf_packed_bytes(&w128));      /* laid out for the next kernel width */
    EXPECT(mi_nerf_packed_bytes(&wide) > 3 * blob && mi_nerf_packed_bytes(&wide) < 4 * blob && mi_nerf_packed_bytes(&too_wide) == 0);   /* ~3.9 x the parameters of 8x256 */
    EXPECT(strstr(mi_nerf_last_error(), "unsupported width W=600") != NULL);

    mi_nerf_render_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.near_ = 2.0f; cfg.far_ = 6.0f; cfg.Sc = 64; cfg.Nf = 128;
    cfg.mode = MI_NERF_MODE_F32;                                                /* ABI 4: the member is called what it is */
#if defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L
    cfg.use_bf16 = MI_NERF_MODE_BF16;                                           /* C11: the ABI 3 name is the same storage (one more version) */
    EXPECT(cfg.mode == MI_NERF_MODE_BF16);
    cfg.mode = MI_NERF_MODE_F32;
#endif
    const size_t ws = mi_nerf_render_workspace_bytes(&cfg, 4096);
    EXPECT(ws > 20u * 1000 * 1000);                                             /* z_c, raw_c, weights_c, z_f, raw_f of 4096 rays */
    mi_nerf_workspace_layout lay;
    EXPECT(mi_nerf_render_workspace_layout(&cfg, 4096, &lay) == MI_NERF_OK && lay.total == ws && lay.raw_f > lay.z_f);

    /* argument errors: status + text, no exception, no GPU touched */
    EXPECT(mi_nerf_render_rays(&net, NULL, NULL, &cfg, NULL, 4096, NULL, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL) == MI_NERF_EINVAL);
    EXPECT(strlen(mi_nerf_last_error()) > 0);
    {   /* a mode that does not exist is refused by name, before any pointer is looked at beyond NULL checks */
        float rays[6] = {0, 0, 0, 0, 0, -1}, out[8];
        char blob_stub[16], ws_stub[16];
        mi_nerf_render_cfg bad = cfg;
        bad.mode = 7; bad.Nf = 0; bad.Sc = 1;
        EXPECT(mi_nerf_render_rays(&net, blob_stub, NULL, &bad, rays, 1, NULL, NULL, ws_stub, (size_t)1 << 20, out, out + 3, NULL, NULL, NULL) == MI_NERF_EINVAL);
        EXPECT(strstr(mi_nerf_last_error(), "MI_NERF_MODE_") != NULL);
        EXPECT(MI_NERF_MODE_F16S == 5 && MI_NERF_MODE_F16S_BF16 == 6 && MI_NERF_MODE_BF16_32 == 3);
    }
    EXPECT(mi_nerf_composite(NULL, NULL, NULL, 6, 4, 64, NULL, NULL, NULL, NULL, NULL, NULL) == MI_NERF_EINVAL);
    void* comm = NULL;
    char id[MI_NERF_COMM_ID_BYTES];
    memset(id, 0, sizeof id);
    EXPECT(mi_nerf_comm_init_rank(id, 8, 8, &comm) == MI_NERF_EINVAL && comm == NULL);
    EXPECT(mi_nerf_all_gather_staging_bytes(8, 378, 504, 4) == (size_t)8 * 48 * 504 * 4 * 4 && mi_nerf_all_gather_staging_bytes(8, 800, 800, 4) == 0);
    EXPECT(mi_nerf_all_gather_tiles(NULL, NULL, 100, 800, 800, 4, NULL, NULL, 0, NULL) == MI_NERF_EINVAL);
    printf("c_abi consumer ok: ABI %d, 8x256 blob %zu bytes, workspace(4096 rays) %zu bytes\n", mi_nerf_abi_version(), blob, ws);
    return 0;
}
