// CONV_FWDB instantiations of the convolution body (own translation unit: parallel builds)
#include "conv_body.h"

HR_DEFINE_CONV_LAUNCH(hr_conv_launch_fwdb, CONV_FWDB)
