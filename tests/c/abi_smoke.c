/* Plain-C consumer of the C ABI (include/amphion_hip.h): the header must compile as C99 and the entry points must link
 * and behave without a GPU (probes, argument validation, error text).  Built and run by tests/test_host_logic.py. */
#include <stdio.h>
#include <string.h>

#include "amphion_hip.h"

int main(void) {
    amp_gen_desc d;
    amp_gen* g = NULL;
    amp_conv* c = NULL;
    amp_mel_desc m;
    int rc;
    memset(&d, 0, sizeof d);
    memset(&m, 0, sizeof m);
    if (amp_version() < 100) { printf("version %d\n", amp_version()); return 1; }
    printf("devices %d\n", amp_device_count());
    /* a descriptor the library must refuse, with a message */
    d.arch = 77;
    rc = amp_gen_create(&d, &g);
    if (rc != AMP_ERR_INVALID || g != NULL || strlen(amp_last_error()) == 0) { printf("bad arch accepted: %d\n", rc); return 2; }
    /* HiFi-GAN V1 descriptor: creation needs no device */
    memset(&d, 0, sizeof d);
    d.arch = AMP_ARCH_HIFIGAN; d.n_in = 80; d.upsample_initial_channel = 512; d.n_stages = 4; d.n_kernels = 3; d.resblock_type = 1;
    { int u[4] = {8, 8, 2, 2}, k[4] = {16, 16, 4, 4}, rk[3] = {3, 7, 11}, i, j;
      for (i = 0; i < 4; ++i) { d.upsample_rates[i] = u[i]; d.upsample_kernel_sizes[i] = k[i]; }
      for (j = 0; j < 3; ++j) { d.resblock_kernel_sizes[j] = rk[j]; d.n_dilations[j] = 3;
                                d.resblock_dilation_sizes[j][0] = 1; d.resblock_dilation_sizes[j][1] = 3; d.resblock_dilation_sizes[j][2] = 5; } }
    rc = amp_gen_create(&d, &g);
    if (rc != AMP_OK || !g) { printf("create failed: %d %s\n", rc, amp_last_error()); return 3; }
    if (amp_gen_hop(g) != 256) { printf("hop %d\n", amp_gen_hop(g)); return 4; }
    /* forward before finalize is a state error, never a crash */
    rc = amp_gen_forward(g, (const float*)&d, NULL, 1, 8, (float*)&d, (void*)&d, 0, NULL);
    if (rc != AMP_ERR_STATE) { printf("forward before finalize: %d\n", rc); return 5; }
    amp_gen_destroy(g);
    /* op-level: NULL arguments and (without a GPU) the no-CPU-fallback refusal */
    if (amp_conv_create(0, 4, 4, 3, 1, 1, 1, NULL, NULL, &c) != AMP_ERR_INVALID) return 6;
    if (amp_conv_create_gated(64, 5, 1, 2, NULL, NULL, &c) != AMP_ERR_INVALID) return 7;
    if (amp_wn_forward(NULL, NULL, 0, NULL, NULL, 0, NULL, 1, 8, NULL, NULL, NULL) != AMP_ERR_INVALID) return 8;
    m.struct_size = (uint32_t)sizeof m;
    m.n_fft = 1024; m.win_size = 1024; m.hop_size = 256; m.n_mel = 80; m.pad_mode = 0;
    if (amp_mel_num_frames(&m, 22016) != 86) { printf("frames %d\n", amp_mel_num_frames(&m, 22016)); return 9; }
    printf("abi ok\n");
    return 0;
}
