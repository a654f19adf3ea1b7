#include "../../mock_mitsuba.h"
