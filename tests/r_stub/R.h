#ifndef RSTUB_R_H
#define RSTUB_R_H
#include <stdlib.h>
#include <string.h>
#include "Rinternals.h"
#endif
