// ABI version probe of libcidnet_hip.so (include/cidnet_hip.h).
#include "cidnet_hip.h"
extern "C" int cidnet_abi_version(void) { return CIDNET_ABI_VERSION; }
