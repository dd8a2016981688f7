// rc_correct_k31.hip -- k_correct compiled for k = 31 over a PACKED table with remainder extension bits (rc_correct_kernel.h)
#include "rc_correct_kernel.h"
RC_K3_SPECIAL(31, 2)
