"""welle.io_amd -- MI355X-native DAB Mode-I PHY backend for welle.io (one hot path).

csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/dabphy.h) -> libdabphy_hip.so
host/      C++ mirror of the reference's backend interface for this path (RadioReceiver facade over the C ABI)
capi.py    ctypes binding used by tests/ and bench.py
synth.py   synthetic Mode-I transmit chain (test / bench input only)
"""
