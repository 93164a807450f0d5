#include "twin.h"

thread_local TwinCtx* g_mcvc_twin = nullptr;
