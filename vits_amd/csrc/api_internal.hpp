#pragma once
// Records a thread-local error message and returns `code` (shared by every entry point).
int vk_fail(int code, const char* msg);
