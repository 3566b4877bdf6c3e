// ca_build_id(): first 16 hex digits of the SHA-1 over the library's sources as built (csrc/Makefile passes it in; engine.source_build_id()
// recomputes it from the tree).  A translation unit of its own: the id depends on every source file, the other objects only on their own.
#include "clonealign_hip.h"
#ifndef CA_BUILD_ID
#define CA_BUILD_ID "unknown"
#endif
extern "C" {
#ifdef CA_LAB
const char* ca_build_id(void) { return "lab-" CA_BUILD_ID; }   // never the tree's id: bench.py and the tests refuse a lab build
#else
const char* ca_build_id(void) { return CA_BUILD_ID; }
#endif
}
