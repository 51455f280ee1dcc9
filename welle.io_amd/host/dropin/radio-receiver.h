// welle.io_amd/host/dropin/radio-receiver.h -- drop-in for the reference's src/backend/radio-receiver.h.
//
// Put this directory in front of the reference's include paths (-Iwelle.io_amd/host/dropin -Iwelle.io_amd/host before
// -Isrc -Isrc/backend) and every `#include "radio-receiver.h"` / `#include "backend/radio-receiver.h"` of welle-cli
// (welle-cli.cpp:53, webradiointerface.cpp:54, tests.cpp:34) resolves here: the class called RadioReceiver is then the
// MI355X facade of gpu_radio_receiver.h -- same constructor, same methods -- and welle-cli's sources build unchanged
// (webradiointerface.h:50 forward-declares `class RadioReceiver;`, so the facade takes the name itself instead of an alias).
// gpu_radio_receiver.cpp is compiled with the same -DGpuRadioReceiver=RadioReceiver; link libdabphy_hip.so instead of
// ofdm-processor / ofdm-decoder / phasereference / fic-handler / msc-handler / dab-audio / viterbi / *-protection / fft objects.
#ifndef DABPHY_DROPIN_RADIO_RECEIVER_H
#define DABPHY_DROPIN_RADIO_RECEIVER_H
#define GpuRadioReceiver RadioReceiver
#include "gpu_radio_receiver.h"
#endif
