// placeholder, filled in below
#include "mp_device.h"
