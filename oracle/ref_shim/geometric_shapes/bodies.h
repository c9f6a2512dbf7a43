#pragma once
#include <geometric_shapes/shapes.h>
namespace bodies { struct Body {}; }
