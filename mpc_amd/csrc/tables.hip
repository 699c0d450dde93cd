// placeholder: device-side table construction
#include "tables.h"
