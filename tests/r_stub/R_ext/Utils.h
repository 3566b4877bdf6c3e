#ifndef RSTUB_UTILS_H
#define RSTUB_UTILS_H
void R_CheckUserInterrupt(void);
#endif
