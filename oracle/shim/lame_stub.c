/* oracle/shim/lame_stub.c -- see lame/lame.h (TEST INFRASTRUCTURE) */
#include "lame/lame.h"
#include <stdlib.h>
struct lame_stub_t { int unused; };
lame_t lame_init(void) { return (lame_t)calloc(1, sizeof(struct lame_stub_t)); }
int lame_set_in_samplerate(lame_t l, int v) { (void)l; (void)v; return 0; }
int lame_set_num_channels(lame_t l, int v) { (void)l; (void)v; return 0; }
int lame_set_VBR(lame_t l, vbr_mode v) { (void)l; (void)v; return 0; }
int lame_set_VBR_q(lame_t l, int v) { (void)l; (void)v; return 0; }
int lame_init_params(lame_t l) { (void)l; return 0; }
int lame_encode_buffer_interleaved(lame_t l, short int pcm[], int n, unsigned char* buf, int size) { (void)l; (void)pcm; (void)n; (void)buf; (void)size; return 0; }
int lame_close(lame_t l) { free(l); return 0; }
