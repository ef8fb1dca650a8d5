// handles.h -- the opaque handle types of include/eesen_hip.h as the library sees them.
#pragma once
#include "net.h"

struct eesen_net : public eesen::Net { using eesen::Net::Net; };
struct eesen_ctc : public eesen::Ctc { using eesen::Ctc::Ctc; };
