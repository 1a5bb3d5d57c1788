// register census of the direct kernel, stage by stage (tools/direct_census.sh); not part of the library
#include "../nvspeechplayer_amd/csrc/klatt_direct.h"
#ifndef CENSUS_MODE
#define CENSUS_MODE 0
#endif
#ifndef CENSUS_CH
#define CENSUS_CH 16
#endif
#ifndef CENSUS_WPE
#define CENSUS_WPE 2
#endif
#ifndef CENSUS_ONLY
#define CENSUS_ONLY 0x7F
#endif
template __global__ void klatt::klatt_direct<CENSUS_MODE, CENSUS_CH, CENSUS_WPE, CENSUS_ONLY>(const klatt::KernelArgs);
