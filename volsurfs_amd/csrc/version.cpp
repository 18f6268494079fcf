// Library identity (checked by the Python loader and the not-gpu symbol test).
#include "../../include/volsurfs_hip.h"
extern "C" int vsa_version(void) { return 1; }
