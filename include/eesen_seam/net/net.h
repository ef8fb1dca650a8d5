/* Shadows the reference header of the same path when include/eesen_seam precedes <eesen>/src on the include path: eesen::Net (src/net/net.h) is
 * provided by the C++ seam over libeesen_hip.so.  See include/eesen_hip_net.h. */
#include "eesen_hip_net.h"
