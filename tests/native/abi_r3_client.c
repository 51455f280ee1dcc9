/* tests/native/abi_r3_client.c -- TEST ONLY: a host object as a maintainer built it against ROUND 3's include/dabphy.h (tests/abi/dabphy_r3.h:
 * dabphy_config without a size member and without decode_shape), linked against TODAY's library.  The exported symbols it calls --
 * dabphy_create, dabphy_get_config -- must still mean that layout (include/dabphy.h "ABI versioning"). */
#include <stdio.h>
#include <string.h>
#include "dabphy_r3.h"

int main(void)
{
    /* the structure sits at the END of a poisoned buffer: a library that read today's (longer) layout through this pointer would pick up
     * the poison behind it as decode_shape and refuse the configuration */
    unsigned char buf[sizeof(dabphy_config) + 64];
    memset(buf, 0x7f, sizeof buf);
    dabphy_config* cfg = (dabphy_config*)buf;
    memset(cfg, 0, sizeof *cfg);
    cfg->n_ensembles = 2; cfg->max_frames = 3; cfg->device = 0; cfg->fft_placement = 1; cfg->freqsync_method = 2; cfg->want_constellation = 1;
    cfg->serial_sync = 1;
    dabphy_handle* h = NULL;
    int r = dabphy_create(cfg, &h);
    if (r != DABPHY_OK || !h) { printf("create failed %d\n", r); return 1; }
    dabphy_config got;
    memset(&got, 0x55, sizeof got);
    r = dabphy_get_config(h, &got);
    if (r != DABPHY_OK) { printf("get_config failed %d\n", r); return 2; }
    printf("ok %u %u %d %d %d %d %d\n", got.n_ensembles, got.max_frames, got.fft_placement, got.freqsync_method, got.want_constellation, got.serial_sync, got.demod_chunk);
    dabphy_destroy(h);
    return 0;
}
