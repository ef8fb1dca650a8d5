/* Shadows the reference header of the same path when include/eesen_seam precedes <eesen>/src on the include path: comm_* (src/net/communicator.h) is
 * provided by the C++ seam over libeesen_hip.so.  See include/eesen_hip_net.h. */
#include "eesen_hip_net.h"
