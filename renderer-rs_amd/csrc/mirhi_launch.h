// mirhi_launch.h -- host-callable launchers of the kernels in mirhi_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "mirhi_device.h"

namespace mirhi {
hipError_t launch_vertex(const PassParams& P, hipStream_t stream);      // no-op unless the scope uses MODEL programs
hipError_t launch_geometry(const PassParams& P, hipStream_t stream);
hipError_t launch_raster(const PassParams& P, uint32_t programs, hipStream_t stream);  // programs: bit0 TRIANGLE, bit1 MODEL / MODEL_FULL, bit2 MODEL_PBR
}  // namespace mirhi
