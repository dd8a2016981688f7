// rc_correct_k23.hip -- k_correct compiled for k = 23 over a PACKED table without remainder extension (rc_correct_kernel.h)
#include "rc_correct_kernel.h"
RC_K3_SPECIAL(23, 1)
