/* tests/harness/acado_caller.c -- TEST TOOL.  Plays the role of the reference's mpc_wrapper.cpp towards
 * libalore_acado_compat.so: it DEFINES the two global structs (mpc_wrapper.cpp:29-30) and exposes their
 * members so that a Python test can drive the acado_* symbols the way MpcWrapper does. */
#include <string.h>

#include "../../include/alore_acado_compat.h"

ACADOvariables acadoVariables;
ACADOworkspace acadoWorkspace;

#define ACC(name, expr) real_t *caller_##name(void) { return (expr); }
ACC(x, acadoVariables.x)
ACC(u, acadoVariables.u)
ACC(od, acadoVariables.od)
ACC(y, acadoVariables.y)
ACC(yN, acadoVariables.yN)
ACC(W, acadoVariables.W)
ACC(WN, acadoVariables.WN)
ACC(x0, acadoVariables.x0)
ACC(lbValues, acadoVariables.lbValues)
ACC(ubValues, acadoVariables.ubValues)
ACC(ws_d, acadoWorkspace.d)
ACC(ws_evGx, acadoWorkspace.evGx)
ACC(ws_evGu, acadoWorkspace.evGu)
ACC(ws_x, acadoWorkspace.x)
ACC(ws_y, acadoWorkspace.y)
ACC(ws_lb, acadoWorkspace.lb)
ACC(ws_ub, acadoWorkspace.ub)
ACC(ws_H, acadoWorkspace.H)
ACC(ws_g, acadoWorkspace.g)
int caller_sizeof_variables(void) { return (int)sizeof(ACADOvariables); }
int caller_sizeof_workspace(void) { return (int)sizeof(ACADOworkspace); }
void caller_reset(void)
{
    memset(&acadoVariables, 0, sizeof(acadoVariables));
    memset(&acadoWorkspace, 0, sizeof(acadoWorkspace));
}
