// rc_correct_k25.hip -- k_correct compiled for k = 25 over a PACKED table without remainder extension (rc_correct_kernel.h)
#include "rc_correct_kernel.h"
RC_K3_SPECIAL(25, 1)
