#pragma once   // TEST-ONLY stub (see README.md)
