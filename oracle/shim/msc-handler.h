// oracle/shim/msc-handler.h -- TEST INFRASTRUCTURE.  See shim/fic-handler.h.
#ifndef SHIM_MSC_HANDLER_H
#define SHIM_MSC_HANDLER_H
#include "fic-handler.h"
class MscHandler {
public:
    SoftbitTap* tap = nullptr;
    void processMscBlock(const softbit_t* data, int16_t blkno) { if (tap && tap->dst) memcpy(tap->dst + 3072 * (blkno - 1), data, 3072); }
};
#endif
