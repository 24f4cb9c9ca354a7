// CONV_FWDS instantiations of the convolution body (forward conv whose input is a residual sum formed in its
// prologue and written out on the side; own translation unit: parallel builds)
#include "conv_body.h"

HR_DEFINE_CONV_LAUNCH(hr_conv_launch_fwds, CONV_FWDS)
