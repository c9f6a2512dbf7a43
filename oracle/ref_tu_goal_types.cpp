// ORACLE SUPPORT: compiles the reference's src/goal_types.cpp, unmodified, from where it lies (BalanceGoal::describe / evaluate; the
// FCL-based TouchGoal is compiled out by the reference's own version guard, as with MoveIt built against FCL >= 0.6)
#include "ref_prelude.h"
#include "goal_types.cpp"
