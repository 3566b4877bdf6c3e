/* stand-in: the registration types live in tests/r_stub/Rinternals.h (test infrastructure, not R's header) */
#include <Rinternals.h>
