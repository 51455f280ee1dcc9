/* oracle/shim/lame/lame.h -- TEST INFRASTRUCTURE: stand-in for <lame/lame.h> so that the reference's welle-cli
 * (welle-cli/webprogrammehandler.cpp:30,108-152: MP3 re-encoding for its web server) links in this image, which has no libmp3lame.
 * Only the web server's audio streaming uses it; the file-decoding path the tests drive (-f file -D) never calls it.
 * The stub encodes nothing: lame_encode_buffer_interleaved reports 0 bytes written. */
#ifndef LAME_STUB_H
#define LAME_STUB_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct lame_stub_t* lame_t;
typedef enum { vbr_off = 0, vbr_default = 4 } vbr_mode;
lame_t lame_init(void);
int lame_set_in_samplerate(lame_t, int);
int lame_set_num_channels(lame_t, int);
int lame_set_VBR(lame_t, vbr_mode);
int lame_set_VBR_q(lame_t, int);
int lame_init_params(lame_t);
int lame_encode_buffer_interleaved(lame_t, short int pcm[], int num_samples, unsigned char* mp3buf, int mp3buf_size);
int lame_close(lame_t);
#ifdef __cplusplus
}
#endif
#endif
