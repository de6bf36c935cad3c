// empty stand-in: minimum_control.h includes it, minimum_control.cpp uses nothing from it (oracle/_ref only)
#pragma once
