#pragma once
#include "frames.hpp"
