// `#include "backend/radio-receiver.h"` (welle-cli.cpp:53, tests.cpp:34) -> the drop-in facade, see ../radio-receiver.h
#include "../radio-receiver.h"
