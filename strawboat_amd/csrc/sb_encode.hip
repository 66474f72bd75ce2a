#include "sb_host.h"
