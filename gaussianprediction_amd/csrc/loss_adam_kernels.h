// loss_adam_kernels.h -- (kernels are private to loss_adam_kernels.hip)
#pragma once
#include "gp_common.h"
