// bio_ik/bio_ik.h — umbrella header, same path as the reference's include/bio_ik/bio_ik.h:35-48
#pragma once
#include "goal.h"
#include "goal_types.h"
