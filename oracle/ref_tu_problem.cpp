// ORACLE SUPPORT: compiles the reference's src/problem.cpp, unmodified, from where it lies
#include "ref_prelude.h"
#include "problem.cpp"
