#pragma once
#include <kdl/frames.hpp>
