// oracle/shim/fic-handler.h -- TEST INFRASTRUCTURE.  Recording stand-in with the two members
// OfdmDecoder (src/backend/ofdm-decoder.cpp:221-228) calls, so that the UNMODIFIED ofdm-decoder.cpp
// can be compiled into oracle/_ref/libwelle_ref_ofdm.so and all 75 x 3072 soft bits of a frame tapped.
#ifndef SHIM_FIC_HANDLER_H
#define SHIM_FIC_HANDLER_H
#include <cstdint>
#include <cstring>
#include "dab-constants.h"
#include "MathHelper.h"
struct SoftbitTap { softbit_t* dst = nullptr; };
class FicHandler {
public:
    SoftbitTap* tap = nullptr;
    void processFicBlock(const softbit_t* data, int16_t blkno) { if (tap && tap->dst) memcpy(tap->dst + 3072 * (blkno - 1), data, 3072); }
};
#endif
