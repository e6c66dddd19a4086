// mirhi_launch.h -- host-callable launchers of the kernels in mirhi_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "mirhi_device.h"

namespace mirhi {
// P: host copy (launch geometry); dev_params: the same parameters in device memory, read by the kernels
hipError_t launch_vertex(const PassParams& P, const PassParams* dev_params, hipStream_t stream);      // no-op unless the scope uses MODEL programs
hipError_t launch_geometry(const PassParams& P, const PassParams* dev_params, hipStream_t stream);
// big_count: the large-triangle counter of this submit's parity (dev_params carries the same pointer)
hipError_t launch_raster(const PassParams& P, const PassParams* dev_params, uint32_t* big_count, uint32_t programs, hipStream_t stream);  // programs: bit0 TRIANGLE, bit1 MODEL / MODEL_FULL, bit2 MODEL_PBR
// sRGB byte -> linear table of the current device (R8G8B8A8_SRGB textures); 256 floats
hipError_t upload_srgb_lut(const float* lut);
}  // namespace mirhi
