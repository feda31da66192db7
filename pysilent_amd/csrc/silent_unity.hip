// The whole library as ONE translation unit: the CPU container's host-only sanitizer build (build.py --host-asan,
// silent_host_shim.h) compiles this; the product is built from the six units separately.
#include "silent_core.hip"
#include "silent_conv_api.hip"
#include "silent_gray_api.hip"
#include "silent_peaks_api.hip"
#include "silent_rgb_api.hip"
#include "silent_pyramid_api.hip"
#include "silent_displayer_api.hip"
