// placeholder for the native codec loop (filled in later this round)
#include "common.h"
